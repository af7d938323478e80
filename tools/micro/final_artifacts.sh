# the round's bench artefacts: the driver's command line, the default line (callers included), the same under rocprofv3
R=$GRAFT_REPO_ROOT
V=${1:-r4_v3}
cd $R
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${V}_bench_driver.json 2> gpurun_out/${V}_bench_driver.err || exit 1
timeout -k 10 500 python bench.py --steps 200 > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err || exit 1
cp gpurun_out/bench_breakdown_n1.json gpurun_out/${V}_event_breakdown.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${V}_prof -- python3 $R/bench.py --steps 100 --no-callers --no-cpu-baseline --no-inference --no-secondary --no-other-configs --no-roofline > $R/gpurun_out/${V}_bench_under_rocprof.json 2> $R/gpurun_out/${V}_bench_under_rocprof.err || exit 1
rm -f $R/gpurun_out/${V}_prof/*/*kernel_trace.csv
echo done
