# A/B of library builds on one workload (GPU box): bash tools/ab_lib.sh <workload> <steps> <lib suffix> ...   ("" = the release library)
wl=$1; steps=$2; shift 2
for v in "$@"; do
  if [ "$v" = "main" ]; then unset OC_HIP_LIB; else export OC_HIP_LIB=$PWD/optimal_crowds_amd/liboc_hip_$v.so; fi
  timeout -k 10 400 python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline --no-work-count > gpurun_out/ab_lib.json 2> gpurun_out/ab_lib.err || { echo "$v failed"; tail -3 gpurun_out/ab_lib.err; continue; }
  python - "$v" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/ab_lib.json') if l.startswith('{')][-1])
b = d['breakdown']; k = b['kernels']; ds = d.get('dense_sweep') or {}
us = lambda x: (1e3 * x) if x else float('nan')
print('%-6s value %7.0f  forces %6.1f us  finish %5.1f us  stage %6.1f us (%s GB/s)  solves %.2f ms / %d   dense: stage %6.1f us solve/key %.2f ms' % (
    sys.argv[1], d['value'], us(k['gcfm_forces']['avg_ms']), us(k['gcfm_finish']['avg_ms']), us(k['hjb_stage']['avg_ms']),
    '%.0f' % (k['hjb_stage']['achieved_GBps'] or 0), 1e3 * b['hjb_solve_s_total_this_rank'], b['hjb_solves_this_rank'],
    us((ds.get('k_stageV') or {}).get('avg_ms')), 1e3 * (ds.get('hjb_solve_s_per_key') or 0)))
PY
done
