"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table
(the same content as `--stats` CSV output): calls, total/avg/min/max duration, % of GPU time."""
import sqlite3
import sys


def summarize(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for n, k, s, a, mn, mx in rows:
        out.append(f"| `{n[:110]}` | {k} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.2f} |")
    return "\n".join(out)


if __name__ == "__main__":
    print(summarize(sys.argv[1]))
