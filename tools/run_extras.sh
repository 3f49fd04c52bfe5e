# The round's other bench lines (GPU box, from the repo root): bash tools/run_extras.sh r03
tag=${1:-r03}
mkdir -p gpurun_out
# the headline workload again with the final bench.py (complete run with the CPU baseline; the driver's command)
python bench.py > gpurun_out/${tag}_bench_plain.log 2> gpurun_out/${tag}_bench_plain.err || tail -3 gpurun_out/${tag}_bench_plain.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_steps20.log 2> gpurun_out/${tag}_bench_steps20.err || tail -3 gpurun_out/${tag}_bench_steps20.err
# float64 HJB state, float32 direction values ('mixed'): what SURVEY 8d's tolerances at full size cost
python bench.py --hjb-dtype mixed --no-cpu-baseline > gpurun_out/${tag}_bench_mixed.log 2> gpurun_out/${tag}_bench_mixed.err || tail -3 gpurun_out/${tag}_bench_mixed.err
python bench.py --hjb-dtype mixed --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_mixed_steps20.log 2>> gpurun_out/${tag}_bench_mixed.err
# every slice written, as the reference does (field = 'slices'): round 2's path on this round's kernels
python bench.py --field slices --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_slices_steps20.log 2> gpurun_out/${tag}_bench_slices.err || tail -3 gpurun_out/${tag}_bench_slices.err
# the reference's update order (level-scheduled on the device): C1 and C2
for wl in room_test exit_opposite; do
  python bench.py --workload $wl --update sequential --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_sequential_$wl.log 2> gpurun_out/${tag}_sequential_$wl.err || tail -3 gpurun_out/${tag}_sequential_$wl.err
  python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_synchronous_$wl.log 2>> gpurun_out/${tag}_sequential_$wl.err
done
# the sharded mode on a one-rank RCCL group (callbacks, then the library's own communicators)
OC_RCCL_NATIVE=0 OC_BENCH_FORCE_SHARDED=1 python bench.py --steps 20 --warmup 2 --mode sharded --no-cpu-baseline --no-ensemble-line > gpurun_out/${tag}_rccl_one_rank.log 2> gpurun_out/${tag}_rccl_one_rank.err || tail -3 gpurun_out/${tag}_rccl_one_rank.err
OC_RCCL_NATIVE=1 OC_BENCH_FORCE_SHARDED=1 python bench.py --steps 20 --warmup 2 --mode sharded --no-cpu-baseline --no-ensemble-line > gpurun_out/${tag}_rccl_native_one_rank.log 2> gpurun_out/${tag}_rccl_native_one_rank.err || tail -3 gpurun_out/${tag}_rccl_native_one_rank.err
# the driver's N = 2 command without a launcher, rehearsed on ONE device over gloo (not a scaling number)
OC_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_rehearsal_n2_one_device.log 2> gpurun_out/${tag}_rehearsal_n2.err || tail -3 gpurun_out/${tag}_rehearsal_n2.err
bash tools/run_workloads.sh $tag
for f in gpurun_out/${tag}_*.log; do echo "$f: $(tail -1 $f | cut -c1-150)"; done
