import sys, os
sys.path.insert(0, '/root/repo')
import rustsasa_amd._capi as c
c.LIB_PATH = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'rustsasa_amd/lib/variants/xcc/librustsasa_amd.so')
import bench
sys.argv = ['bench.py', '--steps', '1', '--warmup', '0', '--cpu-seconds', '0', '--h2h-steps', '0', '--two-steps', '0', '--config5-steps', '0', '--files', '0']
bench.main()
