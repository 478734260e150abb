#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2 'rocpd' sqlite output) into the small text tables committed
under profiles/: per-kernel launch statistics from a --kernel-trace run, per-kernel counter
totals from a --pmc run.

    python tools/rocpd_summary.py trace <results.db>
    python tools/rocpd_summary.py pmc   <results.db> [kernel-substring]
"""
import re
import sqlite3
import sys


def short(name, n=70):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"<.*", "<>", name)
    return name[-n:]


def trace(db):
    c = sqlite3.connect(db).cursor()
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} "
          f"{'%':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'wg':>5s} {'grid':>8s}")
    for r in rows[:15]:
        print(f"{short(r[0]):70s} {r[1]:6d} {r[2] / 1e3:10.1f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f} "
              f"{r[5] / 1e3:9.2f} {100 * r[2] / tot:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:6d} {r[9]:5d} {r[10]:8d}")


def pmc(db, sub=""):
    c = sqlite3.connect(db).cursor()
    rows = c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) "
                     "from counters_collection group by kernel_name, counter_name "
                     "order by sum(value) desc").fetchall()
    print(f"{'kernel':70s} {'counter':>14s} {'launches':>8s} {'sum':>14s} {'avg/launch':>14s} {'avg_us':>9s}")
    for r in rows:
        if sub in r[0]:
            print(f"{short(r[0]):70s} {r[1]:>14s} {r[2]:8d} {r[3]:14.1f} {r[4]:14.2f} {r[5] / 1e3:9.2f}")


if __name__ == "__main__":
    {"trace": trace, "pmc": pmc}[sys.argv[1]](*sys.argv[2:])
