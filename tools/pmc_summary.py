#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md
section HBM prescribes) into profiles/<tag>_pmc_traffic.json: HBM-side bytes per launch and per agent
for the hot kernels.

Correction (gfx950): FETCH_SIZE reports 1/2 of the bytes of coalesced streaming reads. It is calibrated here on
two kernels of this run whose read bytes are known exactly -- adam_polyak_kernel (16 B/lane loads: grads, W,
W_target, m, v = 5 x theta_size x 4 B per agent) and mlp_rows_kernel (4 B/lane loads: one actor per agent) --
and the measured factor is applied to learn_kernel. WRITE_SIZE is exact (checked on adam: 4 x theta_size x 4 B).
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [theta_size]"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0, 0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg[k]
        a[0] += float(row["Counter_Value"]) * 1024.0  # counter unit: KiB
        a[1] += 1
        a[2] += int(row["Grid_Size"]) // int(row["Workgroup_Size"])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    theta = int(sys.argv[4]) if len(sys.argv) > 4 else 76488
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {"source": {"fetch": fetch, "write": write}, "kernels": {}}
    res["fetch_correction"] = {}
    cals = []
    adam = next((k for k in f if k.endswith("adam_polyak_kernel")), None)
    if adam:  # full-slab Adam present (unfused run): grid = (8 blocks, n_sets)
        adam_sets = f[adam][2] / 8
        cals.append((5 * theta * 4 * adam_sets) / f[adam][0])
        res["fetch_correction"]["adam_16B_per_lane"] = cals[-1]
        res["fetch_correction"]["write_check_adam"] = w[adam][0] / (4 * theta * 4 * adam_sets)
    rows = next((k for k in f if "mlp_rows" in k), None)
    if rows:
        actor_bytes = (4 * 256 + 3 * 256 + 256 * 128 + 3 * 128 + 128 + 1 + 2 * 256 + 2 * 128) * 4
        cals.append(actor_bytes * f[rows][2] / f[rows][0])
        res["fetch_correction"]["mlp_rows_4B_per_lane"] = cals[-1]
    # a calibration kernel that did no real work in this run (e.g. mlp_rows_kernel when the conditional actor launch
    # returned at once every step) gives a meaningless factor: keep only plausible ones, else the guide's 2.0
    res["fetch_correction"]["candidates"] = list(cals)
    cals = [c for c in cals if 1.8 <= c <= 2.2]
    corr = sum(cals) / len(cals) if cals else 2.0
    res["fetch_correction"]["applied"] = corr
    res["fetch_correction"]["source"] = "calibrated on kernels of known read bytes" if cals else \
        "MI355X_MICROARCH.md: FETCH_SIZE reports exactly 1/2 of coalesced streaming reads on gfx950"

    for k in f:
        if not k.startswith("avd::"):
            continue
        units = f[k][2] / (8 if k.endswith("adam_polyak_kernel") else (5 if "adam_polyak_ranges" in k else 1))
        fb, wb = f[k][0] * corr, w.get(k, [0, 0, 0])[0]
        res["kernels"][k] = {"launches": f[k][1], "fetch_raw_bytes_per_launch": f[k][0] / f[k][1],
                             "fetch_bytes_per_launch": fb / f[k][1], "write_bytes_per_launch": wb / max(1, w[k][1]),
                             "workgroups_per_launch": f[k][2] / f[k][1],
                             "hbm_bytes_per_unit": (fb + wb) / units if units else None}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["fetch_correction"]))
    for k, v in res["kernels"].items():
        print(f"{k[:60]:60s} launches={v['launches']:3d} HBM bytes/unit={v['hbm_bytes_per_unit']:.0f}")


if __name__ == "__main__":
    main()
