# round 6: SQ occupancy / wait counters of every kernel of an eager step (one pass); the table is built on the box
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-secondary --no-roofline --no-other-configs --no-callers"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc_sq -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_sq.log 2>&1 || exit 1
cd $R
python3 tools/pmc_step_table.py $(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1) > gpurun_out/r06_pmc_sq_step.md
echo done
