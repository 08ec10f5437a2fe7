#!/bin/bash
# quick PMC pass: tools_pmc_quick.sh <tag> "<counters>" [bench args]
set -o pipefail
TAG=$1; CTR=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rays 0 "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,cs in agg.items():
    print(k, {c: f"{v:.4g}" for c,v in cs.items()})
PY
