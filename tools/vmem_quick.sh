# vector-memory counters of one case's trace kernel with the library in VR_LIB_PATH:  bash tools/vmem_quick.sh <out> <case args>
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out; rm -rf $out/vq
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum SQ_INSTS_VALU --output-format csv -d $out/vq -- python3 $GRAFT_REPO_ROOT/tools/case_bench.py "$@" > $out/vq.log 2>&1
python3 - $out/vq <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VMEM_RD": n += 1
print({c: round(v / max(n, 1) / 1e6, 1) for c, v in tot.items()})
PY
