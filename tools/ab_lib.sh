# A/B of whole-library builds on the GPU box: bash tools/ab_lib.sh nt512 nt1024 ... (optimal_crowds_amd/liboc_hip_<tag>.so)
cp optimal_crowds_amd/liboc_hip.so /tmp/keep.so
for t in "$@"; do
  cp optimal_crowds_amd/liboc_hip_$t.so optimal_crowds_amd/liboc_hip.so
  timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/abl_$t.log 2>&1 || { tail -5 gpurun_out/abl_$t.log; exit 1; }
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/abl_%s.log' % t) if l.startswith('{')][-1])
print(t, 'steps/s %.0f' % d['value'], 'hjb total %.4f s' % d['breakdown']['hjb_solve_s_total'],
      {k: round(v['avg_ms'], 4) for k, v in d['breakdown']['kernels'].items() if v['avg_ms']})
PY
done
cp /tmp/keep.so optimal_crowds_amd/liboc_hip.so
