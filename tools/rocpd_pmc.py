"""Aggregate PMC counters per kernel from a rocprofv3 rocpd database (run with --pmc X --kernel-trace).
python tools/rocpd_pmc.py db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*\)$", "", name).replace("void ", "")[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    print("pmc_events columns:", cols)
    kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("kernels columns:", kcols)
    rows = cur.execute("select * from pmc_events limit 3").fetchall()
    for r in rows:
        print(r)
    # best-effort aggregation
    namec = next((c for c in cols if c in ("name", "kernel_name")), None)
    cntc = next((c for c in cols if c in ("counter_name", "pmc_name", "counter", "symbol")), None)
    valc = next((c for c in cols if c in ("value", "counter_value")), None)
    if not (namec and cntc and valc):
        print("could not identify columns; inspect above")
        return
    agg = {}
    for n, c, v in cur.execute(f"select {namec}, {cntc}, {valc} from pmc_events"):
        a = agg.setdefault((short(n), c), [0, 0.0])
        a[0] += 1
        a[1] += float(v)
    lines = ["| kernel | counter | dispatches | sum | per dispatch |", "|---|---|---|---|---|"]
    for (n, c), (k, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {n} | {c} | {k} | {v:.4g} | {v / k:.4g} |")
    out = "\n".join(lines)
    print(out[:6000])
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
