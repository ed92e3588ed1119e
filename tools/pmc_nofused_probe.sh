R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_nf; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
# One-off probe (r04): HBM-side bytes per launch of the UNFUSED nofrl kernels (learn_kernel_l<false>, adam_polyak_kernel), FETCH_SIZE x 2 / WRITE_SIZE.
for cn in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $OUT/$cn -o run -- python3 $R/bench.py --mode nofrl --no-fused --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for cn in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("$OUT/%s/**/*counter_collection.csv"%cn,recursive=True)[0]
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]!=cn: continue
        agg[r["Kernel_Name"].split("(")[0][:70]][0]+=float(r["Counter_Value"])*1024*(2 if cn=="FETCH_SIZE" else 1); agg[r["Kernel_Name"].split("(")[0][:70]][1]+=1
    for k,v in sorted(agg.items(),key=lambda kv:-kv[1][0])[:6]: print(cn,k,v[1],"launches", round(v[0]/v[1]/1e6,1),"MB per launch")
PY
