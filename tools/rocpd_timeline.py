#!/usr/bin/env python
"""Timeline of the LAST bench step in a rocprofv3 rocpd database (kernel-trace): kernel, start offset,
duration and the idle gap since the previous kernel ended (all in microseconds)."""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    ksym = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in ksym else "display_name"
    rows = list(cur.execute(f"select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d "
                            "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
    # one step = from a k_build_cams* launch to the next one
    starts = [i for i, r in enumerate(rows) if "k_build_cams" in r[0]]
    if len(starts) < 3:
        print("not enough steps")
        return
    a, b = starts[-3], starts[-2]
    t0 = rows[a][1]
    prev_end = t0
    busy = 0
    for name, s, e in rows[a:b]:
        m = re.search(r"(k_[a-z_0-9]+)", name)
        short = m.group(1) if m else name.split("(")[0][:50]
        print(f"{short:34s} start {1e-3 * (s - t0):8.1f}  dur {1e-3 * (e - s):7.1f}  gap {1e-3 * (s - prev_end):6.1f}")
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"step span {1e-3 * (prev_end - t0):.1f} us, kernels busy {1e-3 * busy:.1f} us, next step starts at "
          f"{1e-3 * (rows[b][1] - t0):.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
