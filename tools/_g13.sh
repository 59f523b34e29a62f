cd $GRAFT_REPO_ROOT
tools/probe_mix > gpurun_out/g13_probe_mix.log 2>&1; tail -9 gpurun_out/g13_probe_mix.log
timeout 300 python bench.py --weights fp8a --steps 2 --warmup 1 --no-cpu-baseline --no-b4 > gpurun_out/g13_bench_fp8a.json 2> gpurun_out/g13_bench_fp8a.err; head -c 600 gpurun_out/g13_bench_fp8a.json; echo
