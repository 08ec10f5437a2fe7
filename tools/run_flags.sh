cd $GRAFT_REPO_ROOT
for f in 0 1 4 5 8 128; do
VR_DEBUG_FLAGS=$f python3 bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-secondary --no-parity ${1:+--sticking $1} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags $f:', d['value'], 'Mrays/s  trace', d['trace_kernel_ms'], 'gen', d['gen_kernel_ms'])"
done
