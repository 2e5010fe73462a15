#!/bin/bash
# Quick per-kernel timing of one guided-filter pass (8 x 4K, grey and colour src):
#   tools/prof_gf_stats.sh [radius] [lib] [batch]  (through gpurun, from the repo root; prints avg us per
#   kernel, the batch on ONE stream so that the durations are those of kernels running alone)
set -u
export TMPDIR=/tmp
RAD=${1:-45}
LIB=${2:-reflectance_filtering_amd/librf_hip.so}
NB=${3:-8}
O=gpurun_out/gfstats_$(basename $LIB)_r$RAD
rm -rf $O; mkdir -p $O
for kind in grey colour; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$kind -- python3 tools/gf_profile.py $NB 2160 3840 $kind 1 nowall one $RAD $LIB > $O/$kind.log 2>&1
    f=$(find $O/$kind -name "*kernel_stats.csv" | head -1)
    echo "== $kind"
    python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("void rf::") or "gf_" in r["Name"]:
        print("%-60s calls %3s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
