#!/bin/bash
# same-box A/B of the balanced persistent grid (SASPA_GEMM_BALANCE=0 / 1) on the three image sizes of the path
# usage (GPU box, repo root): bash tools/balance_e2e_ab.sh
for r in 1 2; do
  for b in 0 1; do
    echo "round $r SASPA_GEMM_BALANCE=$b"
    SASPA_GEMM_BALANCE=$b python3 tools/nonsquare_bench.py 2>/dev/null | grep "images/s" | cut -c1-60
  done
done
