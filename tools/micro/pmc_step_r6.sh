# round 6: memory-side bytes of every kernel of an eager step (FETCH_SIZE / WRITE_SIZE, separate passes) - and the FETCH pass once
# more with the XCD-local point ranges switched on for the virtual pooling BACKWARD kernels (RL_XCD_BWD=1): what the review asked
# about "why the backward tile kernels lose with the XCD deal".  Tables are built on the box; the raw counter files stay there.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-secondary --no-roofline --no-other-configs --no-callers"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_f.log 2>&1 || exit 1
echo pass F done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_w.log 2>&1 || exit 1
echo pass W done
export RL_XCD_BWD=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fx -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_fx.log 2>&1 || exit 1
unset RL_XCD_BWD
echo pass FX done
cd $R
python3 tools/pmc_hbm_table.py /tmp/pmc_f /tmp/pmc_w 0 > gpurun_out/r06_pmc_hbm_step.md
python3 tools/pmc_hbm_table.py /tmp/pmc_fx /tmp/pmc_w 0 > gpurun_out/r06_pmc_hbm_step_xcd_bwd.md
echo done
