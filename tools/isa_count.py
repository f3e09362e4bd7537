#!/usr/bin/env python3
"""Static instruction counts of one kernel in a -save-temps .s file: tools/isa_count.py file.s mangled-name-substring"""
import re, sys
txt = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
i = next(k for k, l in enumerate(txt) if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].endswith("E"))
j = next(k for k in range(i, len(txt)) if txt[k].strip().startswith(".Lfunc_end"))
body = [l.strip() for l in txt[i:j]]
cnt = {"valu": 0, "mfma": 0, "salu": 0, "branch": 0, "lds": 0, "vmem": 0, "smem": 0, "nop/wait": 0}
for l in body:
    op = l.split(" ")[0]
    if op.startswith("v_mfma"): cnt["mfma"] += 1
    elif op.startswith("v_"): cnt["valu"] += 1
    elif op.startswith("s_cbranch") or op == "s_branch": cnt["branch"] += 1
    elif op in ("s_nop", "s_waitcnt", "s_endpgm", "s_barrier"): cnt["nop/wait"] += 1
    elif op.startswith("s_load") or op.startswith("s_memtime"): cnt["smem"] += 1
    elif op.startswith("s_"): cnt["salu"] += 1
    elif op.startswith("ds_"): cnt["lds"] += 1
    elif op.startswith("global_") or op.startswith("scratch_") or op.startswith("buffer_"): cnt["vmem"] += 1
print(len(body), "lines", cnt)
