R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_infer -- python3 $R/tools/micro/infer_prof.py > $R/gpurun_out/prof_infer.log 2>&1
rm -f $R/gpurun_out/prof_infer/*/*kernel_trace.csv
