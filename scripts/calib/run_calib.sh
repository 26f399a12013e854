#!/bin/bash
# On the GPU box: build the calibration kernels, run them under rocprofv3 with FETCH_SIZE and WRITE_SIZE (separate passes) and
# write gpurun_out/calib/calibration.json = per access pattern, true bytes / counter bytes.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/calib
rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/scripts/calib/fetch_calib.hip -o $O/fetch_calib
cd /tmp && export TMPDIR=/tmp
$O/fetch_calib > $O/true_bytes.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- $O/fetch_calib > /dev/null 2>&1
done
python $R/scripts/calib/make_calibration.py $O
rm -f $O/fetch_calib; find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -delete
