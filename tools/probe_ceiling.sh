#!/bin/bash
# MFMA ceiling of this box with power / clock sampled beside it.  Usage (on the GPU box): tools/probe_ceiling.sh <out.log>
OUT=${1:-gpurun_out/probe_ceiling.log}
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' '; echo; sleep 0.25; done ) > ${OUT}.smi 2>&1 &
POLL=$!
{
  echo "== idle"; rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk"
  echo "== probe_mfma"; timeout 120 tools/probe_mfma 40
  echo "== torch.mm (hipBLASLt)"; timeout 200 python tools/probe_mm_ceiling.py
} > $OUT 2>&1
kill $POLL
# summarise the power / sclk samples: min / median / max over the run
python - "$OUT.smi" >> $OUT <<'PY'
import re, sys, statistics as st
pw, ck = [], []
for line in open(sys.argv[1]):
    m = re.search(r"Power \(W\):\s*([\d.]+)", line) or re.search(r"Power.*?:\s*([\d.]+)", line)
    if m: pw.append(float(m.group(1)))
    m = re.search(r"sclk.*?\((\d+)Mhz\)", line)
    if m: ck.append(int(m.group(1)))
for n, v in (("power W", pw), ("sclk MHz", ck)):
    if v: print(f"smi {n}: n {len(v)} min {min(v)} median {st.median(v)} max {max(v)}")
PY
tail -60 $OUT
