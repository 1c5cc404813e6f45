#!/bin/bash
# Same-box comparison of flash-attention builds that differ in the operand prefetch distance of the v3 loop (SASPA_ATTN_PF):
# the level-0 / level-1 self-attention shapes through tools/attn_bench.py quick, one library per arm (SASPA_HIP_LIB).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for lib in libsaspa_hip.so libsaspa_hip_pf3.so libsaspa_hip_pf4.so; do
  echo "== $lib"
  SASPA_HIP_LIB=$ROOT/saspa-aug_amd/$lib python3 $ROOT/tools/attn_bench.py quick 2>&1 | grep "^B=" | sed 's/v1 .*pre4/pre4/' | cut -c1-160
done
