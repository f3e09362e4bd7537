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
prefix = sys.argv[2] if len(sys.argv) > 2 else "round3_final"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
names = {a: f"{prefix}_{a}" for a in ("bench.json", "bench_under_rocprof.json", "kernel_stats.csv", "pmc.txt",
                                        "bench_uniform1m.json", "single_and_pcie.json", "files_mode.json", "pmc_uniform1m.txt",
                                        "files_mode_1500.json", "files_mode_cif.json", "bench_shard_of_8.json", "two_in_flight.txt", "bench_run2.json")}
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
PRICE = 1.3  # real shader cycles a SIMD spends per vector instruction at this kernel's occupancy: MODELLED, from
             # tools/microbench_clock.hip (dependent v_fma streams on 8 waves per SIMD: 1.30) and from the kernel itself
             # (20 / 40 extra v_mov per atom: +1.15 / +1.3 cycles per atom and SIMD each; 20 s_mov: +1.4): DESIGN 6a


def alu_busy(v, cyc):
    """Vector + matrix pipe occupancy of one dispatch.  Two variants of the same model: the matrix instructions at the
    cycles SQ_VALU_MFMA_BUSY_CYCLES counts, the other vector instructions at PRICE cycles each, over the kernel's
    cycles on the 1024 SIMDs - with and without the cycles in which both kinds were in flight at once
    (SQ_VALU_MFMA_COEXEC_CYCLES).  Neither is a counter; the instruction counts and the busy cycles are."""
    mfma, busy, coexec = v("SQ_INSTS_MFMA"), v("SQ_VALU_MFMA_BUSY_CYCLES"), v("SQ_VALU_MFMA_COEXEC_CYCLES")
    valu = v("SQ_INSTS_VALU")
    other = {k: v(k) for k in ("SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")}
    all_insts = valu + sum(other.values())
    return {
        "modelled": True,
        "price_cycles_per_vector_inst": PRICE,
        "price_source": "tools/microbench_clock.hip (real shader cycles, 8 waves per SIMD) and the in-kernel pad experiment (DESIGN 6a)",
        "mfma_insts_per_launch": mfma, "mfma_busy_cycles_per_launch": busy, "mfma_valu_coexec_cycles_per_launch": coexec,
        "kernel_cycles": round(cyc),
        "frac_sum": round(((valu - mfma) * PRICE + busy) / 1024 / cyc, 3),
        "frac_minus_coexec": round(((valu - mfma) * PRICE + busy - coexec) / 1024 / cyc, 3),
        "insts_all_classes_per_launch": all_insts,
        "issue_frac": round((all_insts * PRICE + busy) / 1024 / cyc, 3),
        "definition": "frac_sum = ((vector - matrix instructions) x price + SQ_VALU_MFMA_BUSY_CYCLES) / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); "
                      "frac_minus_coexec subtracts SQ_VALU_MFMA_COEXEC_CYCLES; issue_frac prices EVERY instruction (vector, scalar, "
                      "branch, LDS, vector memory) at the same cycles - the pad experiment finds a scalar instruction no cheaper "
                      "than a vector one - plus the matrix pipe's busy cycles",
    }


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
        outu["alu_busy"] = alu_busy(vu, vu("GRBM_GUI_ACTIVE") / 8.0)
    except Exception as e:
        print("no alu_busy for the many-point dispatch:", e)
    json.dump(outu, open(os.path.join(dst, "pmc_uniform1m.json"), "w"), indent=2)
    print(json.dumps(outu, indent=1))
