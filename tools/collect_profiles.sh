#!/bin/bash
# Evidence of a build, one gpurun call: rocprofv3 kernel trace + stats of a bench run, the three PMC passes over tools/pmc_step.py
# (separate runs: FETCH_SIZE, WRITE_SIZE, MFMA busy), the FETCH_SIZE calibration per GEMM kernel family, the per-shape table.
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace" && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err || exit 1
echo "== pmc mfma" && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ROOT/tools/pmc_step.py > /dev/null 2> $OUT/pmc_mfma.err || exit 1
echo "== pmc fetch" && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/pmc_step.py > /dev/null 2> $OUT/pmc_fetch.err || exit 1
echo "== pmc write" && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/pmc_step.py > /dev/null 2> $OUT/pmc_write.err || exit 1
echo "== calib fetch" && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_f -- python3 $ROOT/tools/pmc_calib.py > /dev/null 2> $OUT/calib_f.err || exit 1
echo "== calib write" && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_w -- python3 $ROOT/tools/pmc_calib.py > /dev/null 2> $OUT/calib_w.err || exit 1
cd $ROOT
F() { find $OUT/$1 -name "*$2*.csv" | head -1; }
cp "$(F trace kernel_stats)" $OUT/kernel_stats.csv 2>/dev/null
python3 tools/pmc_summary.py "$(F pmc_mfma counter_collection)" > $OUT/pmc_mfma.txt
python3 tools/pmc_summary.py "$(F pmc_fetch counter_collection)" > $OUT/pmc_fetch.txt
python3 tools/pmc_summary.py "$(F pmc_write counter_collection)" > $OUT/pmc_write.txt
python3 tools/pmc_calib.py table "$(F calib_f counter_collection)" "$(F calib_w counter_collection)" $OUT/pmc_calibration.json > $OUT/pmc_calibration.txt
python3 tools/pmc_traffic_json.py "$(F pmc_fetch counter_collection)" "$(F pmc_write counter_collection)" $OUT/pmc_calibration.json > $OUT/pmc_traffic.json
echo "== shapes" && python3 tools/gemm_shapes.py > $OUT/gemm_shapes_512.txt 2>&1
# keep the merged directory small: the raw traces stay on the box
rm -rf $OUT/trace $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write $OUT/calib_f $OUT/calib_w
ls -la $OUT; cat $OUT/pmc_calibration.txt; head -c 600 $OUT/bench_under_trace.json
