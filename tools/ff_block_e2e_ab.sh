#!/bin/bash
# One-launch feed-forward of the level-0 transformer blocks (saspa_ff_block, SASPA_FF_BLOCK=1) against the two launches it replaces:
# same-box A/B through tools/nonsquare_bench.py (512x512, 512x704, 512x768; 3 timed generations each, median), alternating.
# usage (GPU box, repo root): bash tools/ff_block_e2e_ab.sh [rounds = 2]
for r in $(seq 1 ${1:-2}); do
  echo "round $r: two launches (SASPA_FF_BLOCK=0)"
  SASPA_FF_BLOCK=0 python3 tools/nonsquare_bench.py 2>/dev/null | grep "images/s" | cut -c1-60
  echo "round $r: one launch (SASPA_FF_BLOCK=1)"
  SASPA_FF_BLOCK=1 python3 tools/nonsquare_bench.py 2>/dev/null | grep "images/s" | cut -c1-60
done
