#!/bin/bash
# Round 6 dispatch refit (8-wave kernel for the level-1 pointwise layers beside a twin and for the K = 640 GEGLU projection):
# same-box A/B through tools/nonsquare_bench.py (512x512, 512x704, 512x768; 3 timed generations each, median), alternating,
# old gates (SASPA_GEMM_WIDE_KMIN=960 SASPA_GEMM_WIDE_GEGLU_K=1024) against the new defaults.
# usage (GPU box, repo root): bash tools/wide_shortk_e2e_ab.sh [rounds = 2]
for r in $(seq 1 ${1:-2}); do
  echo "round $r: round-5 gates"
  SASPA_GEMM_WIDE_KMIN=960 SASPA_GEMM_WIDE_GEGLU_K=1024 python3 tools/nonsquare_bench.py 2>/dev/null | grep "images/s" | cut -c1-60
  echo "round $r: round-6 gates (defaults)"
  python3 tools/nonsquare_bench.py 2>/dev/null | grep "images/s" | cut -c1-60
done
