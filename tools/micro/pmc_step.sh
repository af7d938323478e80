# memory-side bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of every kernel of an eager training step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-secondary --no-roofline --no-other-configs --no-callers"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_w.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_sq.log 2>&1 || exit 1
grep -o '"steps": [0-9]*' $R/gpurun_out/pmc_f.log | head -1
echo done
