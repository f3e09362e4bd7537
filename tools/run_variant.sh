#!/bin/bash
# run bench.py against a library variant: tools/run_variant.sh NAME [bench args]
V=$1; shift
python3 -c "
import sys; sys.path.insert(0, '.')
import rustsasa_amd._capi as c; c.LIB_PATH = 'rustsasa_amd/lib/variants/$V/librustsasa_amd.so'
import bench; sys.argv = ['bench.py'] + '$*'.split(); bench.main()"
