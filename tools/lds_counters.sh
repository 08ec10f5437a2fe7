# LDS pipe counters of one workload's trace kernel (is the LDS-resident MODE 4 bound by its LDS reads?):
#   bash tools/lds_counters.sh <out dir under gpurun_out> <case args of tools/case_bench.py>
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $out/lds -- python3 $GRAFT_REPO_ROOT/tools/case_bench.py "$@" > $out/lds.log 2>&1
echo "rc=$?"
python3 - $out <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "trace_kernel" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_LDS": n[k] += 1
for k, d in tot.items():
    print(k[:60], "launches", n[k], {c: round(v / max(n[k], 1)) for c, v in d.items()})
PY
