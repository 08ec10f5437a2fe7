# Vector-memory accesses of the trace kernel with parts of the round switched off (VR_DEBUG_FLAGS: 4 no neighbour loop,
# 1 no flux atomics, 8 no walls, 32 no slab packets): what the ~74 scattered lane-loads per segment are made of.
#   bash tools/vmem_by_flag.sh <out dir under gpurun_out> <case args of tools/case_bench.py>
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for f in 0 4 1 5 32; do
  rm -rf $out/vm_$f
  VR_DEBUG_FLAGS=$f rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT --output-format csv -d $out/vm_$f -- python3 $GRAFT_REPO_ROOT/tools/case_bench.py "$@" > $out/vm_$f.log 2>&1
  python3 - $out/vm_$f $f $out/vm_$f.log <<'PY'
import csv, glob, sys, collections, re, json
tot = collections.defaultdict(float); n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VMEM_RD": n += 1
seg = None
for l in open(sys.argv[3]):
    if l.startswith("{"):
        seg = json.loads(l)["segments"]
print("flags", sys.argv[2], "launches", n, "segments", seg, {c: round(v / max(n, 1) / 1e6, 1) for c, v in tot.items()},
      "lane accesses per segment", round(tot["TCP_TOTAL_CACHE_ACCESSES_sum"] / max(n, 1) / seg, 1) if seg else None)
PY
done
