#!/usr/bin/env python3
"""Copies gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<prefix>_* and derives
profiles/pmc_occlusion.json (HBM bytes and VALU instructions per occlusion launch, read by bench.py).
usage: tools/publish_profiles.py <tag> [prefix, default round2_final]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "round6_final"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
names = {a: f"{prefix}_{a}" for a in ("bench.json", "bench_under_rocprof.json", "kernel_stats.csv", "pmc.txt",
                                        "bench_uniform1m.json", "single_and_pcie.json", "files_mode.json", "pmc_uniform1m.txt",
                                        "files_mode_1500.json", "files_mode_cif.json", "bench_shard_of_8.json", "two_in_flight.txt", "bench_run2.json",
                                        "microbench_clock.txt", "h2h_stream.txt", "h2h_stream_trace.txt", "per_call_combined.txt", "files_end_to_end.json",
                                        "files_end_to_end_cif.json")}
for a, b in names.items():
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 10:
        shutil.copy(p, os.path.join(dst, b))
txt = open(os.path.join(src, "pmc.txt")).read()


def val(name):
    return int(re.search(name + r"\s+(\d+)", txt).group(1))


kernel = re.search(r"dispatch \d+: (.*?) grid=", txt).group(1).strip()


sys.path.insert(0, ROOT)
import bench as bench_py  # noqa: E402  (kernel_source_hash: the build these counters belong to)

fetch_kb, write_kb = val("FETCH_SIZE"), val("WRITE_SIZE")
hit, miss = val("TCC_HIT_sum"), val("TCC_MISS_sum")
bench = json.load(open(os.path.join(src, "bench.json")))
out = {
    "kernel": kernel + " (+ k_occlusion_v3 in list mode over the deferred atoms: none in this workload)",
    "workload": "bench.py default (synthetic proteome, 4363 structures, 11.72 M atoms, 100 points)",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum, separate passes "
              "(tools/profile_round.sh, profiles/" + prefix + "_pmc.txt)",
    "FETCH_SIZE_KB": fetch_kb,
    "WRITE_SIZE_KB": write_kb,
    "correction": "gfx950: FETCH_SIZE reports half of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM); the "
                  "sweep/prep loads are 16 B/lane, so the read side is doubled; WRITE_SIZE is exact",
    "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "tcc_hit_rate": round(hit / (hit + miss), 4),
    "valu_insts_per_launch": val("SQ_INSTS_VALU"),
    "salu_insts_per_launch": val("SQ_INSTS_SALU"),
    "kernel_source_sha16": bench_py.kernel_source_hash(),
}
# Prices, validated in round 5 (tools/microbench_clock.hip with HW_ID-observed residency, profiles/round5_microbench_clock.txt):
# a wave64 vector instruction occupies its SIMD-32 for 2 cycles (datasheet: 64 FLOP / clk / SIMD; measured 2.2 over the
# whole span of 7-8 resident waves, 5.2 for one wave alone - latency); a scalar instruction takes one of the CU's scalar
# issue slots, one per cycle and CU (measured: 4.2 cycles per instruction and SIMD with all four SIMDs issuing), so does a
# vector compare that writes an SGPR pair (4.1); round 4's "1.3 cycles per instruction of any kind" divided by an
# assumed residency and implied 1.54 x the datasheet's vector rate.
VECTOR_CYCLES = 2.0


def alu_busy(v, cyc, slots=7):
    """How busy the units of a CU were over one dispatch, from counters and the validated prices above (a MODEL where a
    price is involved: `modelled`): vector ALU = (vector - matrix instructions) x 2 cycles per SIMD; matrix pipe =
    SQ_VALU_MFMA_BUSY_CYCLES per SIMD (a counter), and the two minus the cycles both were in flight
    (SQ_VALU_MFMA_COEXEC_CYCLES); scalar unit = (scalar + branch + scalar-memory instructions) x 1 cycle per CU; LDS pipe =
    SQ_LDS_IDX_ACTIVE per CU (a counter); wave slots = SQ_WAVE_CYCLES x 4 over the cycles of the seven slots per SIMD."""
    mfma, busy, coexec = v("SQ_INSTS_MFMA"), v("SQ_VALU_MFMA_BUSY_CYCLES"), v("SQ_VALU_MFMA_COEXEC_CYCLES")
    valu = v("SQ_INSTS_VALU")

    def opt(name):
        try:
            return v(name)
        except Exception:  # noqa: BLE001  (a pass that was not collected)
            return None
    salu, branch, smem, lds_i = opt("SQ_INSTS_SALU"), opt("SQ_INSTS_BRANCH"), opt("SQ_INSTS_SMEM"), opt("SQ_INSTS_LDS")
    lds_active, wave_cyc, vmem = opt("SQ_LDS_IDX_ACTIVE"), opt("SQ_WAVE_CYCLES"), opt("SQ_INSTS_VMEM_RD")
    out = {
        "modelled": True,
        "vector_cycles_per_inst": VECTOR_CYCLES,
        "price_source": "datasheet rate (64 FLOP / clk / SIMD); tools/microbench_clock.hip with observed residency measures 2.2 "
                        "(profiles/round5_microbench_clock.txt); scalar unit: one instruction per cycle and CU (4.2 cycles per "
                        "instruction and SIMD measured with four SIMDs issuing)",
        "mfma_insts_per_launch": mfma, "mfma_busy_cycles_per_launch": busy, "mfma_valu_coexec_cycles_per_launch": coexec,
        "kernel_cycles": round(cyc),
        "vector_busy": round((valu - mfma) * VECTOR_CYCLES / 1024 / cyc, 3),
        "matrix_busy": round(busy / 1024 / cyc, 3),
        "frac_sum": round(((valu - mfma) * VECTOR_CYCLES + busy) / 1024 / cyc, 3),
        "frac_minus_coexec": round(((valu - mfma) * VECTOR_CYCLES + busy - coexec) / 1024 / cyc, 3),
        "definition": "per SIMD: vector_busy = (vector - matrix instructions) x 2 cycles, matrix_busy = SQ_VALU_MFMA_BUSY_CYCLES, "
                      "frac_sum their sum, frac_minus_coexec without SQ_VALU_MFMA_COEXEC_CYCLES; per CU: scalar_busy = (scalar + "
                      "branch + scalar-memory instructions) x 1 cycle, lds_busy = SQ_LDS_IDX_ACTIVE; all over GRBM_GUI_ACTIVE / 8 of "
                      "the profiled launch; wave_slots_used = SQ_WAVE_CYCLES x 4 / (cycles x 1024 SIMDs x wave_slots_per_simd)",
    }
    if salu is not None and branch is not None:
        out["scalar_busy"] = round((salu + branch + (smem or 0)) / 256 / cyc, 3)
        out["insts_all_classes_per_launch"] = valu + salu + branch + (lds_i or 0) + (vmem or 0)
    if lds_active is not None:
        out["lds_busy"] = round(lds_active / 256 / cyc, 3)
    if wave_cyc is not None:
        out["wave_slots_used"] = round(wave_cyc * 4 / (cyc * 1024 * slots), 3)
        out["wave_slots_per_simd"] = slots
    return out


try:
    out["alu_busy"] = alu_busy(val, val("GRBM_GUI_ACTIVE") / 8.0)
except Exception as e:  # (older pmc.txt without the matrix-pipe pass)
    print("no alu_busy:", e)
json.dump(out, open(os.path.join(dst, "pmc_occlusion.json"), "w"), indent=2)
print(json.dumps(out, indent=1))

# the same for the 1M-atom / 960-point workload (config 5): instruction counts of its occlusion launch
pu = os.path.join(src, "pmc_uniform1m.txt")
if os.path.exists(pu):
    tu = open(pu).read()
    vu = lambda name: int(re.search(name + r"\s+(\d+)", tu).group(1))  # noqa: E731
    outu = {"kernel": re.search(r"dispatch \d+: (.*?) grid=", tu).group(1).strip(),
            "workload": "bench.py --workload uniform1m (1 000 000 atoms in one structure, 960 points)",
            "source": "rocprofv3 --pmc SQ_INSTS_* (tools/profile_round.sh, profiles/" + prefix + "_pmc_uniform1m.txt)",
            "valu_insts_per_launch": vu("SQ_INSTS_VALU"), "mfma_insts_per_launch": vu("SQ_INSTS_MFMA"),
            "salu_insts_per_launch": vu("SQ_INSTS_SALU"), "lds_insts_per_launch": vu("SQ_INSTS_LDS"),
            "kernel_source_sha16": bench_py.kernel_source_hash()}
    try:
        outu["alu_busy"] = alu_busy(vu, vu("GRBM_GUI_ACTIVE") / 8.0, slots=6)  # (80 registers, LDS: six waves per SIMD)
    except Exception as e:
        print("no alu_busy for the many-point dispatch:", e)
    json.dump(outu, open(os.path.join(dst, "pmc_uniform1m.json"), "w"), indent=2)
    print(json.dumps(outu, indent=1))
