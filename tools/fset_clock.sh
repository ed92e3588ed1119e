R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/fset_clk; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -o run -- python3 $R/tools/time_fset.py 4096 5 4 > /dev/null 2>&1
f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("avd::fset::", "")
    if row["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    a = agg[k]; a[0] += float(row["Counter_Value"]); a[1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); a[2] += 1
for k, (c, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if "kernel" in k: print(f"{k[:44]:44s} {ns / n / 1e3:9.1f} us   clock {c / 8 / ns:.3f} GHz")
PY
rm -rf $OUT
