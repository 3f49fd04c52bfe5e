cp optimal_crowds_amd/liboc_hip.so /tmp/keep.so
for n in 0 1 2 3 4; do cp optimal_crowds_amd/liboc_hip_ab$n.so optimal_crowds_amd/liboc_hip.so; timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/ab$n.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/ab$n.log') if l.startswith('{')][-1])
print($n, {k:v['avg_ms'] for k,v in d['breakdown']['kernels'].items()})"; done
