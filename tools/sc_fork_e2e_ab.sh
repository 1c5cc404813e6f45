#!/bin/bash
# Round 6: shortcut convs of the decoder resnets on the side stream (SASPA_FORK_SC=1) against the single-stream order, same box,
# alternating, through tools/nonsquare_bench.py (512x512, 512x704, 512x768).  usage: bash tools/sc_fork_e2e_ab.sh [rounds = 2]
for r in $(seq 1 ${1:-2}); do
  for f in 0 1; do
    echo "round $r SASPA_FORK_SC=$f"
    SASPA_FORK_SC=$f python3 tools/nonsquare_bench.py 2>/dev/null | grep "images/s" | cut -c1-60
  done
done
