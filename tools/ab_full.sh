# full default bench (1000 steps) per tuning string: bash tools/ab_full.sh "resort_every=16" "resort_every=32" ...
for t in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-work-count --tune "$t" > "gpurun_out/abf_$t.log" 2>&1 || { tail -5 "gpurun_out/abf_$t.log"; exit 1; }
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/abf_%s.log' % t) if l.startswith('{')][-1])
print(t, 'steps/s %.0f' % d['value'], 'gcfm ms/step %.4f' % d['breakdown']['gcfm_ms_per_step'],
      {k: round(v['avg_ms'], 4) for k, v in d['breakdown']['kernels'].items() if v['avg_ms']})
PY
done
