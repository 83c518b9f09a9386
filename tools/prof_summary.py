#!/usr/bin/env python3
"""Summarises rocprofv3 (rocpd sqlite) outputs under a directory: per-kernel time stats from the
kernel trace and per-kernel counter sums from each PMC pass.  usage: prof_summary.py <dir>"""
import glob
import os
import sqlite3
import sys


def cols(db, table):
    return [r[1] for r in db.execute(f"pragma table_info({table})")]


def main(root):
    for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        db = sqlite3.connect(path)
        rel = os.path.relpath(path, root)
        print(f"== {rel}")
        kc = cols(db, "kernels")
        name_col = "name" if "name" in kc else [c for c in kc if "name" in c][0]
        rows = db.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start), avg(end-start) "
                          f"from kernels group by {name_col} order by 3 desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        if "trace" in rel:
            print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'pct':>6s}")
            for n, c, s, mn, mx, av in rows[:12]:
                print(f"{n[:60]:60s} {c:6d} {s/1e6:10.3f} {av/1e6:10.4f} {mn/1e6:10.4f} {mx/1e6:10.4f} {100*s/total:6.2f}")
        try:
            cc = cols(db, "counters_collection")
            kn = "kernel_name" if "kernel_name" in cc else name_col
            cn = "counter_name" if "counter_name" in cc else "name"
            vn = "value" if "value" in cc else "counter_value"
            q = db.execute(f"select {kn}, {cn}, count(*), sum({vn}) from counters_collection group by {kn}, {cn} order by 1,2").fetchall()
            for k, c, n, v in q:
                if "path_" in k or "trace_" in k or "wf_" in k:
                    print(f"  {k[:40]:40s} {c:32s} dispatches={n:4d} sum={v:.6g} per_dispatch={v/n:.6g}")
        except sqlite3.OperationalError as e:
            print("  (no counters)", e)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
