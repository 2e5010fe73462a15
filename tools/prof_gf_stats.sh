#!/bin/bash
# Quick per-kernel timing of one guided-filter pass (8 x 4K, grey and colour src):
#   tools/prof_gf_stats.sh   (through gpurun, from the repo root; prints avg ns per kernel)
set -u
export TMPDIR=/tmp
O=gpurun_out/gfstats
rm -rf $O; mkdir -p $O
for kind in grey colour; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$kind -- python3 tools/gf_profile.py 8 2160 3840 $kind > $O/$kind.log 2>&1
    f=$(find $O/$kind -name "*kernel_stats.csv" | head -1)
    echo "== $kind"
    python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("void rf::") or "gf_" in r["Name"]:
        print("%-60s calls %3s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
