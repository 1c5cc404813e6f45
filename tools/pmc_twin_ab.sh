#!/bin/bash
# is the 8-wave conv kernel's MFMA-busy figure diluted by its twin launch under the PMC pass?
# Three rocprofv3 --pmc passes over tools/pmc_step.py: as shipped (two-branch graph, sharing hint), SASPA_TWIN_KS=0
# (two-branch graph, whole-chip dispatch), SASPA_FORK=0 (single-stream graph).  usage: bash tools/pmc_twin_ab.sh <tag>
set -o pipefail
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/twin_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, env assignments...
  local name=$1; shift
  echo "== $name"
  ( export "$@"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/$name -- python3 $ROOT/tools/pmc_step.py > /dev/null 2> $OUT/$name.err ) || return 1
  python3 $ROOT/tools/pmc_summary.py "$(find $OUT/$name -name '*counter_collection*.csv' | head -1)" > $OUT/$name.txt
  rm -rf $OUT/$name
}
run shipped SASPA_DUMMY=1 && run twin_ks0 SASPA_TWIN_KS=0 && run fork0 SASPA_FORK=0
cd $ROOT
head -12 $OUT/shipped.txt $OUT/twin_ks0.txt $OUT/fork0.txt
