#!/bin/bash
# rocprofv3 passes over the C5 shard (128 x 4K, three guided passes) as bench.py runs it:
#   tools/prof_c5_traffic.sh TAG     (through gpurun, from the repo root)
# One --kernel-trace --stats pass, one --pmc FETCH_SIZE and one --pmc WRITE_SIZE pass (separate
# runs, the program directly after `--`), plus the FETCH_SIZE calibration for 1/4/8/12/16-byte
# loads.  tools/make_profiles_c5.py TAG condenses them into profiles/TAG_c5_traffic.{md,json}.
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
B="python3 bench.py --config c5 --steps 2 --warmup 1 --traffic off --cpu-seconds 0 --no-extras"
rm -rf $O/${TAG}_c5_stats $O/${TAG}_c5_fetch $O/${TAG}_c5_write $O/${TAG}_calib
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_c5_stats -- $B > $O/${TAG}_c5_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_c5_fetch -- $B > $O/${TAG}_c5_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_c5_write -- $B > $O/${TAG}_c5_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_calib -- ./tools/microbench/fetch_calib.bin > $O/${TAG}_calib.log 2>&1
tail -2 $O/${TAG}_c5_stats.log
du -sh $O/${TAG}_c5_* $O/${TAG}_calib
