R=$GRAFT_REPO_ROOT; V=${1:-r4_v3}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${V}_prof -- python3 $R/bench.py --steps 100 --no-callers --no-cpu-baseline --no-inference --no-secondary --no-other-configs --no-roofline > $R/gpurun_out/${V}_bench_under_rocprof.json 2> $R/gpurun_out/${V}_bench_under_rocprof.err || exit 1
rm -f $R/gpurun_out/${V}_prof/*/*kernel_trace.csv
echo done
