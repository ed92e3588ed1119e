#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (rocpd sqlite output): prof_top.py <dir-or-db> [n]"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
for db in dbs:
    c = sqlite3.connect(db)
    print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
    for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (n,)):
        print(f"{name[:70]:70s} {calls:6d} {total:12.1f} {avg:10.1f} {pct:6.2f}")
