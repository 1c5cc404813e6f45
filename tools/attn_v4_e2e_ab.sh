#!/bin/bash
# flash_attn_v4_kernel (64 queries per wave, SASPA_ATTN_V4=1, the default) against the v3 loop (SASPA_ATTN_V4=0) end to end:
# same-box A/B through tools/nonsquare_bench.py (512x512, 512x704, 512x768; 3 timed generations each, median), alternating.
# usage (GPU box, repo root): bash tools/attn_v4_e2e_ab.sh [rounds = 2]
for r in $(seq 1 ${1:-2}); do
  echo "round $r: v3 (SASPA_ATTN_V4=0)"
  SASPA_ATTN_V4=0 python3 tools/nonsquare_bench.py 2>/dev/null | grep "images/s" | cut -c1-60
  echo "round $r: v4 (default)"
  python3 tools/nonsquare_bench.py 2>/dev/null | grep "images/s" | cut -c1-60
done
