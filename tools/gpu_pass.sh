#!/bin/bash
# One GPU-box pass (run from the repo root on the box): usage tools/gpu_pass.sh <tag> <steps...>
#   bench     the default bench line (B = 1 headline + the b4 point + cpu_baseline)
#   b4prof    rocprofv3 kernel summary of one num_images=4 image
#   b1prof    rocprofv3 kernel summary of one num_images=1 image
#   pmc / pmc512   PMC passes over the GEMM shapes at 128 / 512 rows
#   tests     the full -m gpu suite;  t:<expr>  pytest -m gpu -k <expr>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=$1; shift
mkdir -p $O
cd $R
for step in "$@"; do
  case $step in
    bench) timeout 400 python bench.py --steps 3 --warmup 1 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -c 1500 $O/${TAG}_bench_default.json ;;
    b4prof|b1prof|b8prof|b16prof)
      NI=${step#b}; NI=${NI%prof}
      (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$step && timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$step -o p -- python $R/bench.py --num-images $NI --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-b4 --no-throughput > $O/${TAG}_rocprof_$step.log 2>&1)
      DB=$(ls /tmp/prof_$step/*.db /tmp/prof_$step/*/*.db 2>/dev/null | head -1)
      [ -n "$DB" ] && python tools/rocpd_stats.py $DB $O/${TAG}_kernel_stats_$step.md > /dev/null 2>&1; head -30 $O/${TAG}_kernel_stats_$step.md ;;
    pmc) timeout 240 bash tools/run_pmc_passes.sh $O/pmc > $O/${TAG}_pmc.log 2>&1; cp $O/pmc/pmc_gemm_traffic.json $O/${TAG}_pmc_gemm_traffic.json; tail -3 $O/${TAG}_pmc.log ;;
    pmc512) BD_PMC_ROWS=512 timeout 300 bash tools/run_pmc_passes.sh $O/pmc512 > $O/${TAG}_pmc512.log 2>&1; cp $O/pmc512/pmc_gemm_traffic.json $O/${TAG}_pmc_gemm_traffic_rows512.json; tail -3 $O/${TAG}_pmc512.log ;;
    tests) timeout 700 python -m pytest tests -m gpu -q -x > $O/${TAG}_pytest_full.log 2>&1; tail -5 $O/${TAG}_pytest_full.log ;;
    t:*) timeout 600 python -m pytest tests -m gpu -q -x -k "${step#t:}" > $O/${TAG}_pytest_sel.log 2>&1; tail -15 $O/${TAG}_pytest_sel.log ;;
    sh:*) timeout 600 bash -c "${step#sh:}" > $O/${TAG}_sh.log 2>&1; tail -40 $O/${TAG}_sh.log ;;
  esac
done
