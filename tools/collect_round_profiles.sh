#!/bin/bash
# Everything behind profiles/<TAG>_* in one go (through gpurun, from the repo root):
#   gpurun --timeout 3000 -- bash tools/collect_round_profiles.sh r06
# then copy the summaries from gpurun_out/ into profiles/ (see the last lines).
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
# ONE long fuzz run against the oracle on the final library of the round
RF_FUZZ_SECONDS=${RF_FUZZ_SECONDS:-150} RF_FUZZ_SEED=${RF_FUZZ_SEED:-6000} python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s > $O/${TAG}_fuzz_final.txt 2>&1; tail -3 $O/${TAG}_fuzz_final.txt
tools/prof_bench.sh > $O/${TAG}_prof_bench.log 2>&1; python tools/make_profiles.py $TAG > $O/${TAG}_make.log 2>&1
tools/prof_c5_traffic.sh $TAG > $O/${TAG}_prof_c5.log 2>&1; python tools/make_profiles_c5.py $TAG > $O/${TAG}_make_c5.log 2>&1
tools/prof_gf_cnn.sh $TAG 8 > /dev/null 2>&1; python tools/make_profiles_gf.py $TAG 8 > $O/${TAG}_make_gf.log 2>&1
python tools/bench_other.py > $O/${TAG}_bench_other.json 2> $O/${TAG}_bench_other.err
python tools/gf_c5_exp.py --rounds 3 --check --out $O/${TAG}_c5_switches.json base gf_no_compact=1 gf_one_stream=1 gf_s1_legacy_strips=1 gf_exact=1 gf_exact=1,gf_exact_all_flagged=1 gf_exp_skip=6 gf_exp_skip=1 > /dev/null 2>&1
python tools/gf_c5_exp.py --rounds 3 --src colour --batch 64 --check --out $O/${TAG}_c5_colour.json base gf_no_compact=1 gf_cw_chan_run=1 gf_one_stream=1 gf_exact=1 gf_exp_skip=6 gf_exp_skip=1 > /dev/null 2>&1
python tools/fuzz_parity.py --seconds 60 --seed 5 > $O/${TAG}_fuzz_parity.json 2> $O/${TAG}_fuzz.err
python tools/stress_sizes.py > $O/${TAG}_stress_sizes.json 2> $O/${TAG}_stress.err
# the summaries make_profiles*.py wrote into profiles/ on this box; -n: never over a fresh output above
cp -n profiles/${TAG}_* profiles/jbf_pmc_traffic.json $O/ 2>/dev/null
# gpurun copies at most 64 MiB back: the raw rocprofv3 trees have been condensed above
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls $O | grep ${TAG} | head -60
