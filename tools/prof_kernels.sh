#!/bin/bash
# Per-kernel durations of one C5 run on ONE stream (kernels alone, no co-running halves):
#   tools/prof_kernels.sh TAG [RF_DEBUG_OPTIONS value]      (through gpurun, from the repo root)
# writes gpurun_out/TAG_kernels.txt: kernel, calls, total ms, average ms (library kernels only).
set -u
TAG=${1:-k}
OPTS=${2:-}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -rf $O/${TAG}_kst
RF_DEBUG_OPTIONS="gf_one_stream=1${OPTS:+,$OPTS}" rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kst -- \
    python3 bench.py --config c5 --steps 2 --warmup 1 --traffic off --cpu-seconds 0 --no-extras > $O/${TAG}_kst.log 2>&1
python3 - "$O/${TAG}_kst" > $O/${TAG}_kernels.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "rf::" in n:
            short = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("rf::", "").split("(")[0]
            print("%-44s calls %3s  total %8.2f ms  avg %7.3f ms" % (short, r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
cat $O/${TAG}_kernels.txt
rm -rf $O/${TAG}_kst
