#!/usr/bin/env python
"""Turns a rocprofv3 rocpd sqlite database (default output of `rocprofv3 --kernel-trace --stats`) into a small text summary
(per-kernel calls / total / average duration in microseconds) that can be committed under profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None, top=12):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["kernel,calls,total_us,avg_us,percent"]
    for name, calls, total, avg, pct in rows[:top]:
        short = name.split("(")[0] if not name.startswith("void at::") else "torch::" + name.split("<")[0].split("::")[-1]
        lines.append(f"{short},{calls},{total:.1f},{avg:.2f},{pct:.2f}")
    text = "\n".join(lines) + "\n"
    if out_path:
        with open(out_path, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
