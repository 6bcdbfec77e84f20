#!/bin/bash
# usage (through gpurun): bash tools/pmc_calibrate.sh <tag>  -> profiles/<tag>_pmc_calibration.json (also copied to gpurun_out/)
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_cal_fetch -o p -- python3 $R/tools/pmc_calibrate.py > $OUT/${TAG}_cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_cal_write -o p -- python3 $R/tools/pmc_calibrate.py > $OUT/${TAG}_cal_write.log 2>&1
python3 $R/tools/pmc_calibrate.py summarize $TAG
cp $R/profiles/${TAG}_pmc_calibration.json $OUT/
rm -rf $OUT/${TAG}_cal_fetch $OUT/${TAG}_cal_write
