"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table (like --stats CSV).
python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.md]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("void ", "")
    if name.startswith("at::native::") or name.startswith("void at::"):
        name = "torch::" + name.split("::")[-1][:60]
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    t0 = min(r[1] for r in rows)
    t1 = max(r[2] for r in rows)
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = [f"# rocprofv3 kernel-trace summary: {len(rows)} dispatches, {tot / 1e6:.1f} ms kernel time, "
             f"{(t1 - t0) / 1e6:.1f} ms first-to-last", "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {n} | {a[0]} | {a[1] / 1e6:.2f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | {100 * a[1] / tot:.1f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
