set -u
export TMPDIR=/tmp
O=gpurun_out
RF_FUZZ_SECONDS=30 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s > $O/r03_fuzz_oracle.txt 2>&1; tail -3 $O/r03_fuzz_oracle.txt
tools/prof_bench.sh > $O/r03_prof_bench.log 2>&1; python tools/make_profiles.py r03 > $O/r03_make.log 2>&1
tools/prof_gf_cnn.sh r03 8 > /dev/null 2>&1; python tools/make_profiles_gf.py r03 8 > $O/r03_make_gf.log 2>&1
for k in grey colour; do tools/prof_gf_sq.sh r03 $k 8; done
python tools/bench_other.py > $O/r03_bench_other.json 2> $O/r03_bench_other.err
python tools/gf_radius_ab.py --radii 8,20,30,45,52,60,96 --out $O/r03_gf_radius.json > $O/r03_gf_radius.log 2>&1
python3 tools/gf_stream_sweep.py > $O/r03_gf_stream_sweep.txt 2>/dev/null
python tools/fuzz_parity.py --seconds 150 --seed 3 > $O/r03_fuzz_parity.json 2> $O/r03_fuzz.err
python tools/stress_sizes.py > $O/r03_stress_sizes.json 2> $O/r03_stress.err
python3 tools/gf_stamp_run.py 45 8 grey > $O/r03_gf_stamps_grey.json 2>/dev/null
python3 tools/gf_stamp_run.py 45 8 colour > $O/r03_gf_stamps_colour.json 2>/dev/null
python3 tools/cnn_cmp.py > $O/r03_cnn_cmp.txt 2>&1
# the summaries make_profiles*.py wrote into profiles/ on this box; -n: never over a fresh output above
cp -n profiles/r03_* profiles/jbf_pmc_traffic.json $O/ 2>/dev/null
ls $O | grep r03 | head -50
