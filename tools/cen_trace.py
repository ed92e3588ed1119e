#!/usr/bin/env python3
"""Timeline of one centralized update from a rocprofv3 --kernel-trace csv: start / end of the learn chunks and of their Adam + Polyak
passes relative to the first launch (do they overlap?). usage: cen_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "learn_kernel_c" in r["Kernel_Name"] or "adam_polyak_r" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = rows[-n:]  # the last update(s)
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = "learn" if "learn_kernel_c" in r["Kernel_Name"] else "adam "
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{name} queue {r.get('Queue_Id', '?'):>3} start {s / 1e3:9.1f} us  end {e / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?')}")
