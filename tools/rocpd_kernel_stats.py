#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table
(count, total, average, min, max in microseconds) -- the same content as `--stats` CSV output."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    ksym = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in ksym else ("display_name" if "display_name" in ksym else ksym[1])
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "group by s.id order by 3 desc")
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    lines = ["name,calls,total_us,avg_us,min_us,max_us,percent"]
    for name, n, tot, mn, mx in rows:
        short = name.split("(")[0][:110]
        lines.append(f'"{short}",{n},{tot / 1e3:.2f},{tot / 1e3 / n:.3f},{mn / 1e3:.3f},{mx / 1e3:.3f},{100.0 * tot / total:.2f}')
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
