#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into the text summary kept under profiles/."""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w") as f:
        f.write(f"# {title}\n# rocprofv3 --kernel-trace --stats (durations in microseconds)\n")
        f.write(f"{'calls':>8} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel\n")
        for name, calls, total, avg, pct in rows:
            name = name if len(name) <= 200 else name[:197] + "..."  # rocPRIM template names run to kilobytes
            f.write(f"{calls:>8} {total:>14.3f} {avg:>12.3f} {pct:>7.2f}  {name}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
