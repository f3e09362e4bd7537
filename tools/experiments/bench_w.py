"""bench.py with another lane count W of the reference's SIMD (rsasa_context_set_simd_width) for every context it makes:
python tools/experiments/bench_w.py W [bench args]   (GPU box).  100 points: W = 8 or 16 leave 4 remainder points, W = 4 none."""
import os, sys
sys.path.insert(0, os.getcwd())
import rustsasa_amd.engine as e
W = int(sys.argv[1])
_init = e.Context.__init__
def init(self, device=0, simd_width=W):
    _init(self, device, simd_width)
e.Context.__init__ = init
import bench
sys.argv = ["bench.py"] + sys.argv[2:]
bench.main()
