#!/bin/bash
# The rocprofv3 passes behind profiles/<TAG>_bench*.{json,csv,md} and jbf_pmc_traffic.json
# (run through gpurun from the repo root; tools/make_profiles.py condenses the output):
#   tools/prof_bench.sh
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_calib $O/prof_sq $O/prof_grbm $O/prof_sq_colour $O/prof_grbm_colour
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
B="python3 bench.py --cpu-seconds 0 --no-extras --traffic off"   # never nest a profiler under a profiler
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- $B --steps 5 --warmup 1 > $O/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- $B --steps 2 --warmup 1 > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- $B --steps 2 --warmup 1 > $O/prof_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_calib -- ./tools/microbench/fetch_calib.bin > $O/prof_calib.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/prof_sq -- $B --batch 32 --steps 1 --warmup 1 > $O/prof_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/prof_grbm -- $B --batch 32 --steps 1 --warmup 1 > $O/prof_grbm.log 2>&1
# the same two passes with a 3-channel colour src (the colour tap loop, 44 VALU instructions per column step)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/prof_sq_colour -- $B --src colour --batch 32 --steps 1 --warmup 1 > $O/prof_sq_colour.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/prof_grbm_colour -- $B --src colour --batch 32 --steps 1 --warmup 1 > $O/prof_grbm_colour.log 2>&1
ls $O
