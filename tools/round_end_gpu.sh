#!/bin/bash
# One GPU-box pass for the numbers a round commits under profiles/ (run from the repo root on the box):
#   PMC passes over the GEMM shapes -> profiles/r04_pmc_gemm_traffic.json (bench.py's roofline.traffic reads it), the default
#   bench line, the full -m gpu suite, and the rocprofv3 kernel summary of the bench command at 1024 px.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 180 bash tools/run_pmc_passes.sh > $O/c6_pmc.log 2>&1 && cp $O/pmc/pmc_gemm_traffic.json profiles/r04_pmc_gemm_traffic.json && cp $O/pmc/pmc_gemm_traffic.json $O/c6_pmc_gemm_traffic.json
timeout 200 python bench.py > $O/c6_bench_default.json 2> $O/c6_bench_default.err
timeout 640 python -m pytest tests -m gpu -q > $O/c6_pytest_full.log 2>&1
timeout 150 python bench.py --num-images 4 --steps 1 --warmup 1 --no-cpu-baseline > $O/c6_bench_b4.json 2> $O/c6_bench_b4.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && timeout 170 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof -o p -- python $R/bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline > $O/c6_rocprof_bench.log 2>&1)
DB=$(ls /tmp/prof/*.db /tmp/prof/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB $O/c6_kernel_stats_1024px.md > /dev/null 2>&1
tail -3 $O/c6_pytest_full.log
tail -c 600 $O/c6_bench_default.json
tail -5 $O/c6_pmc.log
