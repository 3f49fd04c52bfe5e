# A/B of a launch-shape knob on the C4 step (GPU box): bash tools/ab_tune.sh "<knob>=a" "<knob>=b" ...   [STEPS=300]
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null; mkdir -p gpurun_out
for v in "$@"; do
  timeout -k 10 300 python bench.py --steps ${STEPS:-300} --warmup 20 --no-cpu-baseline --no-work-count --no-dense-line ${WL:+--workload $WL} --tune "$v" > gpurun_out/ab_tune.json 2> gpurun_out/ab_tune.err || { echo "$v failed"; tail -3 gpurun_out/ab_tune.err; continue; }
  python - "$v" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/ab_tune.json') if l.startswith('{')][-1])
k = d['breakdown']['kernels']
print('%-24s value %7.0f  forces %6.1f us  finish %5.1f us  solves %.2f ms' % (sys.argv[1], d['value'], 1e3 * k['gcfm_forces']['avg_ms'], 1e3 * k['gcfm_finish']['avg_ms'], 1e3 * d['breakdown']['hjb_solve_s_total_this_rank']))
PY
done
