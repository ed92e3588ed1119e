#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of tools/pmc_workload.py) into HBM-side bytes per launch
for every avd:: kernel, with the counters CALIBRATED on the 1 GiB stream copy of the same pass: factor = 2^30 / counted bytes of the
copy kernel (expected 2.0 +- 0.05 for FETCH_SIZE on gfx950, 1.0 for WRITE_SIZE). usage: pmc_traffic.py fetch.csv write.csv out.json"""
import collections
import csv
import json
import sys

GIB = float(1 << 30)


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = row["Kernel_Name"]
        agg[k][0] += float(row["Counter_Value"]) * 1024.0  # counter unit: KiB
        agg[k][1] += 1
    return agg


def calib(agg):
    """the copy kernel: the launches whose counted bytes are the largest non-avd ones (three identical 1 GiB copies)"""
    cands = {k: v for k, v in agg.items() if "avd::" not in k and "fw::" not in k and "elementwise" in k.lower() and v[1] == 3}  # launched exactly three times
    if not cands:
        return None, None
    k = max(cands, key=lambda k: cands[k][0] / cands[k][1])
    return k, GIB / (cands[k][0] / cands[k][1])


fetch, write, out = sys.argv[1:4]
f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
kf, cf = calib(f)
kw, cw = calib(w)
res = {"calibration": {"copy_kernel": kf, "fetch_factor": cf, "write_factor": cw,
                       "note": "factor = 2^30 bytes / counted bytes per launch of a 1 GiB torch copy in the SAME pass; applied below"},
       "kernels": {}}
cf_, cw_ = (cf or 2.0), (cw or 1.0)
for k in sorted(f, key=lambda k: -f[k][0]):
    if "avd::" not in k and "fw::" not in k:  # (fw:: = the persistent kernels of csrc/wide.hip)
        continue
    name = k.split("(")[0].replace("void ", "")
    res["kernels"][name] = {"launches": f[k][1], "fetch_bytes_per_launch": cf_ * f[k][0] / f[k][1],
                            "write_bytes_per_launch": cw_ * w.get(k, [0, 1])[0] / max(1, w.get(k, [0, 1])[1])}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["calibration"]))
for k, v in list(res["kernels"].items())[:14]:
    print(f"{k[:70]:70s} launches {v['launches']:3d}  fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:9.1f} MB")
