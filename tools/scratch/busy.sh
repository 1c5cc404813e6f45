cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/busy; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1
cd $R && python tools/busy_summary.py $O/tr > $O/busy.txt; cat $O/busy.txt; rm -rf $O/tr
