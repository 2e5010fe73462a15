#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd $R; mkdir -p gpurun_out/r05
timeout 900 python tools/gf_c5_exp.py --batch 64 --rounds 3 --out gpurun_out/r05/exp5_matrix.json \
  base gf_seg_rows=135 gf_seg_rows=270 gf_stagger=1,gf_seg_rows=135 gf_stagger=1,gf_seg_rows=270 \
  gf_exp_skip=6,gf_seg_rows=135 gf_exp_skip=6,gf_seg_rows=270 \
  gf_exp_skip=4,gf_seg_rows=135 gf_exp_skip=4,gf_stagger=1,gf_seg_rows=135 gf_exp_skip=4,gf_stagger=1,gf_seg_rows=270 \
  gf_exp_skip=2,gf_seg_rows=135 gf_exp_skip=2,gf_stagger=1,gf_seg_rows=135 gf_exp_skip=2,gf_stagger=1,gf_seg_rows=270 \
  > gpurun_out/r05/exp5_matrix.log 2>&1
