#!/usr/bin/env python
"""Per-kernel PMC averages from a rocprofv3 rocpd database (--pmc run)."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    pmc_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")]
    info_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")]
    disp_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    print("pmc_event cols:", pmc_cols, file=sys.stderr)
    print("info_pmc cols:", info_cols, file=sys.stderr)
    print("dispatch cols:", disp_cols, file=sys.stderr)
    name_col = "name" if "name" in info_cols else "symbol"
    # a raw SQ / TCP / TCC counter comes as one row per hardware instance (XCD x shader engine ...) and dispatch:
    # SUM the instances of a dispatch, then average over the dispatches of a kernel
    q = (f"select kname, cname, count(*), avg(v), sum(v) from ("
         f" select s.kernel_name as kname, s.id as sid, p.{name_col} as cname, d.id as did, sum(e.value) as v"
         "  from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id"
         "  join rocpd_kernel_dispatch d on e.event_id = d.event_id"
         "  join rocpd_info_kernel_symbol s on d.kernel_id = s.id"
         f"  group by s.id, p.{name_col}, d.id)"
         " group by sid, cname order by kname")
    print("kernel,counter,dispatches,avg_per_dispatch,sum")
    for name, ctr, n, avg, tot in cur.execute(q):
        print(f'"{name.split("(")[0][:60]}",{ctr},{n},{avg:.1f},{tot:.0f}')


if __name__ == "__main__":
    main(sys.argv[1])
