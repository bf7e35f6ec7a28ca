#!/usr/bin/env python
"""Timeline of one STEADY-STATE bench step (k_build_pairs to the next k_build_pairs, the step at the lower quartile of the spans of
the back-to-back ones) from a rocprofv3 rocpd database: kernel, start offset, duration, idle gap since the previous kernel
ended (microseconds); memory copies of the same stream are listed too when the database has them."""
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
    starts = [i for i, r in enumerate(rows) if "k_build_pairs" in r[0]]
    spans = [(rows[starts[k + 1]][1] - rows[starts[k]][1], k) for k in range(len(starts) - 1)]
    tight = sorted(s for s in spans if s[0] < 2e6)  # < 2 ms between launches: the timed loop
    if not tight:
        print("no back-to-back steps")
        return
    # the lower quartile: bench.py's extra steps with per-kernel events (LT_FINE_TIMERS=2) and its steps with the tail are
    # back to back too, and slower than the timed ones
    span, k = tight[len(tight) // 4]
    a, b = starts[k], starts[k + 1]
    t0 = rows[a][1]
    prev_end = t0
    busy = 0
    for name, s, e in rows[a:b]:
        m = re.search(r"(k_[a-z_0-9]+)", name)
        short = m.group(1) if m else name.split("(")[0][:50]
        print(f"{short:34s} start {1e-3 * (s - t0):8.1f}  dur {1e-3 * (e - s):7.1f}  gap {1e-3 * (s - prev_end):6.1f}")
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"step: {1e-3 * span:.1f} us from launch to launch ({len(tight)} back-to-back steps, lower quartile shown), "
          f"kernels busy {1e-3 * busy:.1f} us, idle {1e-3 * (span - busy):.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
