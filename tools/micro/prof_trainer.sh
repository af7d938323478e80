cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RL_PIPELINE_RNG=device
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trainer -- python3 $R/tools/trainer_profile.py > $R/gpurun_out/prof_trainer.log 2>&1
