#!/usr/bin/env python
"""Timeline of the sigmaenv kernels from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d <dir> -o <name> -- ...`): name, queue,
stream, start and duration of the last launches -- shows how kernels of different streams (env shards, the actor) overlap."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
rows = list(db.execute(f"select k.kernel_name, d.queue_id, d.stream_id, d.start, d.end, d.grid_size_x from {kd} d join {ks} k on d.kernel_id=k.id order by d.start"))
rows = [r for r in rows if "sigmaenv" in r[0]][-int(sys.argv[2]) if len(sys.argv) > 2 else -24:]
t0 = rows[0][3]
for r in rows:
    print(r[0][:30].ljust(30), "queue", r[1], "stream", r[2], "start %9.1f us  dur %6.1f us  grid %d" % ((r[3] - t0) / 1e3, (r[4] - r[3]) / 1e3, r[5]))
