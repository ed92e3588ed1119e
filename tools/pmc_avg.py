#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel: pmc_avg.py <counter_collection.csv> <COUNTER> [out.json]"""
import collections
import csv
import json
import sys

path, counter = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(path)):
    if row["Counter_Name"] != counter:
        continue
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][0] += float(row["Counter_Value"])
    agg[k][1] += 1
res = {k: {"launches": n, "mean_" + counter: s / n} for k, (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]) if k.startswith(("avd::", "fw::"))}
for k, v in res.items():
    print(f"{k[:70]:70s} launches={v['launches']:4d} {counter}={v['mean_' + counter]:.2f}")
if len(sys.argv) > 3:
    json.dump({"counter": counter, "source": path, "kernels": res}, open(sys.argv[3], "w"), indent=1)
