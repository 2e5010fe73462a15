#!/bin/bash
# rocprofv3 passes for the guided-filter and CNN kernels (run through gpurun from the repo root):
#   tools/prof_gf_cnn.sh TAG [GF_BATCH]
# Writes gpurun_out/TAG_{gf,gfc,cnn}_{stats,fetch,write}/ ; tools/make_profiles_gf.py condenses them.
# Each counter pass is its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass); the program
# follows `--` directly (no env/bash hop under the profiler).
set -u
TAG=${1:-r02}
NB=${2:-8}
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
run() {  # name, pmc-or-empty, program...
    local name=$1 pmc=$2
    shift 2
    if [ -z "$pmc" ]; then
        rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_${name}_stats" -- "$@" \
            > "$OUT/${TAG}_${name}_stats.log" 2>&1
    else
        rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$OUT/${TAG}_${name}_${pmc%% *}" -- "$@" \
            > "$OUT/${TAG}_${name}_${pmc%% *}.log" 2>&1
    fi
}
for pmc in "" FETCH_SIZE WRITE_SIZE; do
    WALL=$([ -z "$pmc" ] && echo wall || echo nowall)  # counter passes serialise the kernels
    run gf "$pmc" python3 tools/gf_profile.py "$NB" 2160 3840 grey 1 $WALL
    run gfc "$pmc" python3 tools/gf_profile.py "$NB" 2160 3840 colour 1 $WALL
    run cnn "$pmc" python3 tools/cnn_profile.py 256
    # event timing of the same launches as the --stats pass (counter passes serialise and slow the kernels)
    [ -z "$pmc" ] && cp "$OUT/cnn_profile_events.json" "$OUT/${TAG}_cnn_events.json"
done
ls "$OUT"
