# usage: tools/pmc_case.sh <tag> <program args...>   (PMC passes over one command, summary of trace/gen kernels)
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $out/a -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $out/b -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_FLAT --output-format csv -d $out/c -- python3 "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.defaultdict(set)
for f in glob.glob("$out/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-40:]
        if "trace_kernel" in k or "gen_kernel" in k:
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
for k,cs in agg.items():
    print(k, "dispatches/pass", len(cnt[k])/3)
    for c,v in sorted(cs.items()): print("   %-24s %.4g"%(c,v))
PY
