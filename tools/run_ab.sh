cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 5 --warmup 1 --cpu-rays 0 --no-secondary --no-parity "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'Mrays/s  trace', d['trace_kernel_ms'], 'gen', d['gen_kernel_ms'])"; }
for i in 1 2; do
echo -n "base s=1: "; run
echo -n "nopq s=1: "; VR_DEBUG_FLAGS=128 run
echo -n "base s=.1: "; run --sticking 0.1
echo -n "mode0 s=.1: "; VR_GENERAL_FLAT=0 run --sticking 0.1
echo -n "nopq s=.1: "; VR_DEBUG_FLAGS=128 run --sticking 0.1
done
