/* Plain C99 consumer of include/rustsasa_amd.h: the header must compile as C and the
 * library must link and behave without any C++/HIP types on the caller's side.
 * Exit code 0 = ok on a GPU host, 0 with "no device" printed on a GPU-less host. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rustsasa_amd.h"

int main(void)
{
    if (rsasa_abi_version() != RSASA_ABI_VERSION) return 10;
    if (sizeof(rsasa_atom_t) != 24) return 11;
    int n_dev = -1;
    if (rsasa_device_count(&n_dev) != RSASA_OK) return 12;
    float sx[100], sy[100], sz[100];
    if (rsasa_sphere_points(100, sx, sy, sz) != RSASA_OK || sz[0] != 1.0f) return 13;

    rsasa_context_t *ctx = NULL;
    int rc = rsasa_context_create(0, &ctx);
    if (n_dev == 0) {
        if (rc != RSASA_ERR_NO_DEVICE || ctx != NULL) return 14;
        printf("no device: %s\n", rsasa_status_string(rc));
        return 0;
    }
    if (rc != RSASA_OK) return 15;

    /* two overlapping spheres (reference tests/sanity.rs:64-85) + one far away */
    rsasa_atom_t atoms[3];
    memset(atoms, 0, sizeof atoms);
    atoms[0].radius = atoms[1].radius = atoms[2].radius = 2.0f;
    atoms[1].position[0] = 4.0f;
    atoms[2].position[0] = 40.0f;
    atoms[0].id = 1; atoms[1].id = 2; atoms[2].id = 3;
    float out[3] = {-1.f, -1.f, -1.f};
    rc = rsasa_calculate_sasa_internal(ctx, atoms, 3, 1.4f, 5000, -1, out);
    if (rc != RSASA_OK) { printf("%s\n", rsasa_context_last_error(ctx)); return 16; }
    const double pi = 3.14159265358979323846, r = 3.4;
    const double full = 4 * pi * r * r, exposed = full - 2 * pi * r * (r - 2.0);
    if (out[0] < exposed * 0.99 || out[0] > exposed * 1.01) return 17;
    if (out[1] < exposed * 0.99 || out[1] > exposed * 1.01) return 18;
    if (out[2] < full * 0.999 || out[2] > full * 1.001) return 19;
    /* ABI 2: lane count round trip, NUMA binding of the calling thread (node -1: nothing to bind), wait-all with
     * nothing in flight */
    int w = 0, node = -2;
    if (rsasa_context_set_simd_width(ctx, 4) != RSASA_OK || rsasa_context_get_simd_width(ctx, &w) != RSASA_OK || w != 4) return 23;
    if (rsasa_context_set_simd_width(ctx, 3) != RSASA_ERR_INVALID_ARGUMENT) return 24;
    if (rsasa_context_set_simd_width(ctx, 8) != RSASA_OK) return 25;
    if (rsasa_context_bind_thread(ctx, &node) != RSASA_OK || node < -1) return 26;
    if (rsasa_batch_wait_all(ctx) != RSASA_OK || rsasa_batch_wait(ctx) != RSASA_OK) return 27;
    /* ABI 3: a second context with the first one's settings; a stream of host batches: two queued, waited for in order,
     * a wait with nothing queued returns at once */
    {
        rsasa_context_t *other = NULL;
        int w2 = 0;
        if (rsasa_context_create(0, &other) != RSASA_OK) return 28;
        if (rsasa_context_set_simd_width(ctx, 16) != RSASA_OK || rsasa_context_clone_settings(other, ctx) != RSASA_OK ||
            rsasa_context_get_simd_width(other, &w2) != RSASA_OK || w2 != 16) return 29;
        if (rsasa_context_set_simd_width(ctx, 8) != RSASA_OK || rsasa_context_destroy(other) != RSASA_OK) return 30;
        float x[3] = {0.f, 4.f, 40.f}, y[3] = {0.f, 0.f, 0.f}, z[3] = {0.f, 0.f, 0.f}, rad[3] = {2.f, 2.f, 2.f};
        uint32_t so[2] = {0u, 3u};
        float a1[3] = {-1.f, -1.f, -1.f}, a2[3] = {-1.f, -1.f, -1.f};
        if (rsasa_host_batch_wait(ctx) != RSASA_OK) return 31;
        if (rsasa_host_batch_enqueue(ctx, x, y, z, rad, NULL, so, 1, 1.4f, 5000, a1, NULL, 0, NULL) != RSASA_OK) return 32;
        if (rsasa_host_batch_enqueue(ctx, x, y, z, rad, NULL, so, 1, 1.4f, 5000, a2, NULL, 0, NULL) != RSASA_OK) return 33;
        if (rsasa_host_batch_wait(ctx) != RSASA_OK || a1[0] != out[0] || a1[2] != out[2]) return 34;
        if (rsasa_host_batch_wait_all(ctx) != RSASA_OK || a2[1] != out[1]) return 35;
        uint64_t dropped = 99;
        if (rsasa_context_ids_dropped(ctx, &dropped) != RSASA_OK || dropped != 0) return 36;  /* (small batches are not checked) */
    }
    /* empty input is valid and touches nothing */
    if (rsasa_calculate_sasa_internal(ctx, NULL, 0, 1.4f, 100, 1, NULL) != RSASA_OK) return 20;
    /* invalid arguments are reported, not crashed on */
    if (rsasa_calculate_sasa_internal(ctx, atoms, 3, 1.4f, 0, 1, out) != RSASA_ERR_INVALID_ARGUMENT) return 21;
    if (rsasa_context_destroy(ctx) != RSASA_OK) return 22;
    printf("abi ok: %.3f %.3f %.3f\n", out[0], out[1], out[2]);
    return 0;
}
