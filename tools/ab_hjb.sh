# A/B of the HJB stage-launch shapes on the GPU box: bash tools/ab_hjb.sh [workload]
wl=${1:-metro_station}
for f in ${FUSE:-0 1}; do
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-work-count --no-dense-line --tune ${KNOB:-hjb_fuse}=$f > gpurun_out/ab_hjb_${wl}_$f.json 2> gpurun_out/ab_hjb_${wl}_$f.err || { echo "fuse $f failed"; tail -3 gpurun_out/ab_hjb_${wl}_$f.err; continue; }
  python - $f gpurun_out/ab_hjb_${wl}_$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
b = d["breakdown"]; k = b["kernels"]
print("fuse=%s value %.1f  solve_s %.5f  stage %.4f ms (%.0f GB/s)  out %.4f ms (%.0f GB/s) attempts %s" % (
    sys.argv[1], d["value"], b["hjb_solve_s_total_this_rank"] / max(b["hjb_solves_this_rank"], 1), k["hjb_stage"]["avg_ms"],
    k["hjb_stage"]["achieved_GBps"], k["hjb_out"]["avg_ms"], k["hjb_out"]["achieved_GBps"], b["rk_attempts"]))
PY
done
