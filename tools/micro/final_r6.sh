# the round's committed evidence in one call: default line, driver-form line, rocprofv3 kernel stats of the metric's configuration,
# PMC traffic of the dominant function's largest shape
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err || exit 1
cp gpurun_out/bench_breakdown_n1.json gpurun_out/r06_event_breakdown.json
echo default line done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver.json 2> gpurun_out/r06_bench_driver.err || exit 1
echo driver line done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r6 -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-inference --no-other-configs --no-callers --steps 80 --warmup 10 > $R/gpurun_out/r06_bench_under_rocprof.json 2> $R/gpurun_out/r06_rocprof.err || exit 1
cp $(find /tmp/prof_r6 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06_rocprofv3_kernel_stats.csv
echo rocprof done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_gf -- python3 $R/tools/pmc_kernel.py gemm 81920 256 256 > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_gw -- python3 $R/tools/pmc_kernel.py gemm 81920 256 256 > /dev/null 2>&1 || exit 1
cd $R
python3 - <<'PY' > gpurun_out/r06_pmc_wgemm2_81920x256x256.txt
import csv, glob
for d, c in (("/tmp/pmc_gf", "FETCH_SIZE"), ("/tmp/pmc_gw", "WRITE_SIZE")):
    vals = {}
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == c and "wgemm2_kernel" in r["Kernel_Name"]:
                vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    v = list(vals.values())
    print(c, "launches", len(v), "mean", sum(v) / max(len(v), 1), "min", min(v), "max", max(v))
PY
cat gpurun_out/r06_pmc_wgemm2_81920x256x256.txt
echo done
