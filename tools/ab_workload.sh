# full bench of one workload per tuning string: bash tools/ab_workload.sh open_room "prep_blocks=200" "prep_blocks=256"
w=$1; shift
for t in "$@"; do
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-work-count --workload $w --tune "$t" > "gpurun_out/abw_${w}_$t.log" 2>&1 || { tail -5 "gpurun_out/abw_${w}_$t.log"; exit 1; }
  python - "$w" "$t" <<'PY'
import json, sys
w, t = sys.argv[1], sys.argv[2]
d = json.loads([l for l in open('gpurun_out/abw_%s_%s.log' % (w, t)) if l.startswith('{')][-1])
print(w, t, 'steps/s %.0f' % d['value'], {k: round(v['avg_ms'], 4) for k, v in d['breakdown']['kernels'].items() if v['avg_ms']})
PY
done
