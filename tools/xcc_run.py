#!/usr/bin/env python3
"""One bench step on the -DMX_DEBUG_XCC build of the library (make OUT=../lib/variants/xcc/... EXTRA=-DMX_DEBUG_XCC):
k_occlusion_mx prints the XCD (HW_REG_XCC_ID) a few of its workgroups run on.  On the GPU box:
    python tools/xcc_run.py 2>&1 | grep MXXCC"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustsasa_amd._capi as c  # noqa: E402

c.LIB_PATH = os.path.join(ROOT, "rustsasa_amd", "lib", "variants", "xcc", "librustsasa_amd.so")
import bench  # noqa: E402

sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--h2h-steps", "0", "--two-steps", "0",
            "--config5-steps", "0", "--files", "0", "--real-steps", "0"]
bench.main()
