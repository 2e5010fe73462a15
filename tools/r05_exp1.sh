#!/bin/bash
# round-5 experiment 1: staggered schedule / stage-1 occupancy cap at the C5 shard
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gf" 2>&1 | tail -3 > gpurun_out/r05/exp1_pytest.txt
timeout 900 python tools/gf_c5_exp.py --check --rounds 3 --out gpurun_out/r05/exp1_matrix.json \
  base gf_stagger=1 gf_stagger=1,gf_parts=4 gf_stagger=1,gf_s1_cap=2 gf_stagger=1,gf_s1_cap=3 \
  gf_stagger=1,gf_s1_cap=2,gf_s1_min_wgs=512 gf_stagger=1,gf_s1_cap=2,gf_parts=4 gf_s1_cap=2 gf_s1_cap=3 \
  gf_stagger=1,gf_seg_rows=270 gf_stagger=1,gf_seg_rows=1080 gf_stagger=1,gf_parts=8 \
  > gpurun_out/r05/exp1_matrix.log 2>&1
cd /tmp
for v in aligned:"" stag:"gf_stagger=1" stagcap2:"gf_stagger=1,gf_s1_cap=2"; do
  name=${v%%:*}; opts=${v#*:}
  RF_DEBUG_OPTIONS="$opts" timeout 600 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/r05/ovl_$name" -o t -- \
    python3 "$GRAFT_REPO_ROOT/bench.py" --config c5 --steps 2 --warmup 1 --traffic off --cpu-seconds 0 --no-extras \
    > "$GRAFT_REPO_ROOT/gpurun_out/r05/ovl_$name.log" 2>&1
  python3 "$GRAFT_REPO_ROOT/tools/gf_overlap.py" "$GRAFT_REPO_ROOT/gpurun_out/r05/ovl_$name" --label "$name ($opts)" \
    > "$GRAFT_REPO_ROOT/gpurun_out/r05/ovl_$name.md" 2>&1
  # keep only the small csv
  find "$GRAFT_REPO_ROOT/gpurun_out/r05/ovl_$name" -name "*.db" -delete
done
