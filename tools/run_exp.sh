cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
N=$(echo $C | tr ' ' '_' | cut -c1-30)
rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/expc/$N -- python3 $R/tools/case_bench.py trench3d 0.1 4000 1 > /dev/null 2>&1
echo "pmc $C rc=$?"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(float)
for f in glob.glob("$R/gpurun_out/expc/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"])
for c,v in sorted(agg.items()): print("%-36s %.4g"%(c,v))
PY
