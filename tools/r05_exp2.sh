#!/bin/bash
# round-5 experiment 2: kernel-trace timelines of the schedules
R="$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r05
cd /tmp
for v in aligned:"" stag:"gf_stagger=1" stagcap2:"gf_stagger=1,gf_s1_cap=2" stagcap3:"gf_stagger=1,gf_s1_cap=3" one:"gf_one_stream=1"; do
  name=${v%%:*}; opts=${v#*:}
  rm -rf $R/gpurun_out/r05/ovl_$name
  RF_DEBUG_OPTIONS="$opts" timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/r05/ovl_$name" -o t -- \
    python3 "$R/bench.py" --config c5 --steps 2 --warmup 1 --traffic off --cpu-seconds 0 --no-extras \
    > "$R/gpurun_out/r05/ovl_$name.log" 2>&1
  python3 "$R/tools/gf_overlap.py" "$R/gpurun_out/r05/ovl_$name" --label "$name ($opts)" \
    > "$R/gpurun_out/r05/ovl_$name.md" 2>&1
done
