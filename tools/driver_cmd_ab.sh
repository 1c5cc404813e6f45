#!/bin/bash
# Same-box A/B at the DRIVER's bench command (`bench.py --gpus 1 --steps 20 --warmup 5`): the round-4 final tree
# (`git archive 0da8daf` extracted to _ab/r4 and built there) against this tree, alternating, then this tree with the
# round-5 defaults switched back one at a time (SASPA_GEMM_PP_LOOP=0, SASPA_GEMM_BALANCE=0).  `--no-cpu-baseline`: the
# CPU leg is outside the timed region and only costs box time.
# usage: tools/driver_cmd_ab.sh [alternations = 3] [knobs = 1]   -> gpurun_out/r6_ab/*.json + summary.txt
set -u
N=${1:-3}
KNOBS=${2:-1}
OUT=gpurun_out/r6_ab
mkdir -p $OUT
run() {   # tag, tree, env...
  tag=$1; tree=$2; shift 2
  ( cd $tree && env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ) > $OUT/$tag.json 2> $OUT/$tag.err
  python3 - "$tag" $OUT/$tag.json <<'EOF' | tee -a gpurun_out/r6_ab/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:28s} {j['value']:.4f} images/s  ms_per_step {j['ms_per_step']:.2f}  clock {j.get('clock')}")
except Exception as e:
    print(f"{sys.argv[1]:28s} FAILED {e}")
EOF
}
: > $OUT/summary.txt
for i in $(seq 1 $N); do
  run r4_$i _ab/r4 X=1
  run r5_$i . X=1
done
if [ "$KNOBS" = "1" ]; then
  run r5_pploop0 . SASPA_GEMM_PP_LOOP=0
  run r5_balance0 . SASPA_GEMM_BALANCE=0
  run r5_again . X=1
fi
cat $OUT/summary.txt
