#!/bin/bash
# PMC passes over the GEMM shapes of one AR step (run on the GPU box from the repo root): one rocprofv3 --pmc pass per counter
# group, kernel trace only (never combined with sys/hip/hsa tracing), rocpd output; then parse into profiles-style JSON.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$ROOT/gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=${pass%%:*}; ctrs=${pass#*:}
  rm -rf "$OUT/$tag"
  rocprofv3 --pmc $ctrs --kernel-trace --output-format rocpd -d "$OUT/$tag" -o p -- python "$ROOT/tools/pmc_gemm_traffic.py" run > "$OUT/$tag.log" 2>&1 || echo "pass $tag failed (see $OUT/$tag.log)"
done
cd "$ROOT"
db() { ls $OUT/$1/*.db $OUT/$1/*/*.db 2>/dev/null | head -1; }
python tools/pmc_gemm_traffic.py parse "$OUT/pmc_gemm_traffic.json" "$(db fetch)" "$(db write)" "$(db sq)"
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/sq"
