# The profile runs of profiles/README.md in one go (on the GPU box, from the repo root): bash tools/run_profiles.sh r02 [workload]
# Order: the PMC passes first, condensed on the box (tools/collect_profiles.py writes profiles/<tag>_pmc_*.json with the
# kernel sources' hash), so that the bench lines taken afterwards carry roofline.traffic from passes over the same sources.
tag=${1:-r02}
wl=${2:-metro_station}
sfx=""; [ "$wl" != "metro_station" ] && sfx="_$wl"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}${sfx}_pmc_fetch -- python bench.py --workload $wl --steps 20 --warmup 2 --no-cpu-baseline --no-work-count --no-dense-line > gpurun_out/${tag}${sfx}_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}${sfx}_pmc_write -- python bench.py --workload $wl --steps 20 --warmup 2 --no-cpu-baseline --no-work-count --no-dense-line > gpurun_out/${tag}${sfx}_pmc_write.log 2>&1 || exit 1
# SQ counters of the step kernels and of the HJB kernels, per workload (each file is stamped with the workload it was taken on)
bash tools/pmc_step.sh ${tag}${sfx} --workload $wl > gpurun_out/${tag}${sfx}_pmc_step.txt 2>&1
KERNELS="k_stageV|k_out|k_control|k_list|k_density" bash tools/pmc_step.sh ${tag}${sfx}_hjb --workload $wl > gpurun_out/${tag}${sfx}_pmc_hjb.txt 2>&1
python tools/collect_profiles.py $tag $wl > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}${sfx}_stats -- python bench.py --workload $wl --no-work-count --no-dense-line --no-cpu-baseline > gpurun_out/${tag}${sfx}_bench.log 2>&1 || exit 1
python bench.py --workload $wl > gpurun_out/${tag}${sfx}_bench_plain.log 2>&1 || exit 1
python bench.py --workload $wl --steps 20 --warmup 5 > gpurun_out/${tag}${sfx}_bench_steps20.log 2>&1 || exit 1
tail -1 gpurun_out/${tag}${sfx}_bench_plain.log | cut -c1-400
