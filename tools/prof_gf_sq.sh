#!/bin/bash
# SQ / GRBM counters of the guided-filter kernels (separate passes; program directly after `--`).
#   tools/prof_gf_sq.sh TAG [grey|colour] [batch]
set -u
TAG=${1:-r02}
KIND=${2:-grey}
NB=${3:-8}
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d "$OUT/${TAG}_gfsq_${KIND}" -- python3 tools/gf_profile.py "$NB" 2160 3840 "$KIND" > "$OUT/${TAG}_gfsq_${KIND}.log" 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU \
    --kernel-trace --output-format csv -d "$OUT/${TAG}_gfsq2_${KIND}" -- python3 tools/gf_profile.py "$NB" 2160 3840 "$KIND" > "$OUT/${TAG}_gfsq2_${KIND}.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/${TAG}_gfgrbm_${KIND}" -- python3 tools/gf_profile.py "$NB" 2160 3840 "$KIND" > "$OUT/${TAG}_gfgrbm_${KIND}.log" 2>&1
