#!/bin/bash
# usage (inside gpurun): tools/r04_phase.sh <tag> <debug values...>  -- DEV-build phase report per LGNN_FUSED_DEBUG value
tag=$1; shift
mkdir -p gpurun_out
for d in "$@"; do
  LGNN_FUSED_DEBUG=$d LGNN_LIB_DIR=lib_dev LGNN_PHASE_REPORT=1 timeout -k 10 300 python tools/phase_report.py arxiv 2 > gpurun_out/${tag}_phase$d.log 2>&1
  echo "debug $d"; tail -8 gpurun_out/${tag}_phase$d.log | cut -c1-130
done
