# A/B of launch-shape knobs on the GPU box: bash tools/ab_tune.sh "pair_split=1" "pair_split=2" ...
# an entry may carry extra bench flags after a space: "pair_split=2 --wall-radius 0.01"
for t in "$@"; do
  set -- $t
  tune=$1; shift
  tag=$(echo "$t" | tr ' /' '__')
  timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --tune "$tune" "$@" > "gpurun_out/ab_$tag.log" 2>&1 || { tail -5 "gpurun_out/ab_$tag.log"; exit 1; }
  python - "$tag" <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/ab_%s.log' % t) if l.startswith('{')][-1])
print(t, 'steps/s %.0f' % d['value'], 'gcfm kernel ms/step %.4f' % (d['breakdown']['gcfm_kernel_ms_per_step'] or 0),
      {k: round(v['avg_ms'], 4) for k, v in d['breakdown']['kernels'].items() if v['avg_ms']})
PY
done
