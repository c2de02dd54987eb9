#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table (like --stats csv).
usage: tools/rocpd_stats.py results.db [skip_first_n_dispatch_fraction]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[-110:] if ' g=' in name else name[:110]


def main(path, out=None, by_grid=False):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    if by_grid and gcol:
        rows = cur.execute(f"select name || ' g=' || {gcol}, start, end from kernels").fetchall()
    else:
        rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = [f"{'kernel':110s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:110s} {a[0]:7d} {a[1]:12.1f} {a[1] / a[0]:10.2f} {a[2]:9.2f} {a[3]:9.2f} {100 * a[1] / tot:6.2f}")
    lines.append(f"TOTAL kernel time {tot:.1f} us over {len(rows)} dispatches")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--by-grid"]
    main(args[0], args[1] if len(args) > 1 else None, "--by-grid" in sys.argv)
