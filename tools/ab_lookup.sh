# needs the diagnostic build: make -C optimal_crowds_amd/csrc diag   (the release library rejects the "diag" knob)
export OC_HIP_LIB=$PWD/optimal_crowds_amd/liboc_hip_diag.so
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null; mkdir -p gpurun_out
for v in "ring:--field ring" "ringnolook:--field ring --tune diag=8" "slices:--field slices" "slicesnolook:--field slices --tune diag=8"; do
  name=${v%%:*}; args=${v#*:}
  timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-work-count --no-dense-line $args > gpurun_out/r03e_$name.json 2> gpurun_out/r03e_$name.err
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r03e_$name.json') if l.startswith('{')][-1])
k=d['breakdown']['kernels']
print('$name', 'value %.0f' % d['value'], 'forces %.1f us finish %.1f us' % (1e3*k['gcfm_forces']['avg_ms'], 1e3*k['gcfm_finish']['avg_ms']), 'solve total %.2f ms' % (1e3*d['breakdown']['hjb_solve_s_total_this_rank']))
PY
done
