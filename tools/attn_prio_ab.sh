#!/bin/bash
# A/B of a static s_setprio for waves 4-7 of the 8-wave flash attention (built on the box; one library per value)
# usage (GPU box, repo root): bash tools/attn_prio_ab.sh
set -e
cd saspa-aug_amd/csrc
for v in 1 2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -mllvm -amdgpu-mfma-vgpr-form=1 -DSASPA_ATTN_YOUNG_PRIO=$v -c saspa_attn.hip -o /tmp/attn_prio$v.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v "saspa_attn.o\|abl\|\.ct") /tmp/attn_prio$v.o -o ../libsaspa_hip_prio$v.so
done
cd ../..
for r in 1 2; do
  echo "== round $r: shipped"; python3 tools/attn_bench.py 2>/dev/null | grep -i "4096\|1024" | head -4
  for v in 1 2; do echo "== round $r: young half at priority $v"; SASPA_HIP_LIB=$PWD/saspa-aug_amd/libsaspa_hip_prio$v.so python3 tools/attn_bench.py 2>/dev/null | grep -i "4096\|1024" | head -4; done
done
rm -f saspa-aug_amd/libsaspa_hip_prio*.so
