cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in old new; do
  if [ $v = old ]; then export RL_HIP_LIB=$R/3d_recognizer_amd/csrc/librandla_hip_old.so; else unset RL_HIP_LIB; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wg_$v -- python3 $R/tools/pmc_kernel.py wgradb 81920 256 256 6 > $R/gpurun_out/pmc_wg_$v.log 2>&1 || exit 1
done
