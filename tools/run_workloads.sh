# Bench lines of the other BASELINE workloads (single GPU): bash tools/run_workloads.sh r02
tag=${1:-r02}
for wl in exit_opposite slalom room_test; do
  python bench.py --workload $wl > gpurun_out/${tag}_wl_$wl.json 2> gpurun_out/${tag}_wl_$wl.err || { echo "$wl failed"; tail -3 gpurun_out/${tag}_wl_$wl.err; }
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_wl_${wl}_steps20.json 2>> gpurun_out/${tag}_wl_$wl.err
  cut -c1-220 gpurun_out/${tag}_wl_$wl.json
done
