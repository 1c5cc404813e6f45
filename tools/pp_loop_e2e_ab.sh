#!/bin/bash
# same-box A/B of the 8-wave kernel's K loop flavours through the bench: SASPA_GEMM_PP_LOOP=0 (four 20-MFMA phases per K-tile)
# against =2 (two 40-MFMA intervals), alternating, 3 rounds.   usage (GPU box, repo root): bash tools/pp_loop_e2e_ab.sh
for r in 1 2 3; do
  for l in 0 2; do
    v=$(SASPA_GEMM_PP_LOOP=$l python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['value'])")
    echo "round $r SASPA_GEMM_PP_LOOP=$l  $v images/s"
  done
done
