#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace run (rocpd sqlite output).

    python tools/kstats.py gpurun_out/prof_x/serial_results.db [out.csv]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    name = re.sub(r"^void ", "", name).replace("fm::", "")
    return name[:80]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0, min(end-start)/1000.0, "
                      "max(end-start)/1000.0 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,percent"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f"\"{short(n)}\",{c},{t:.1f},{a:.2f},{mn:.2f},{mx:.2f},{100 * t / tot:.2f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
