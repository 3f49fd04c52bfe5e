# The profile runs of profiles/README.md in one go (on the GPU box, from the repo root): bash tools/run_profiles.sh r01
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python bench.py > gpurun_out/${tag}_bench_plain.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python bench.py --no-work-count > gpurun_out/${tag}_bench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_pmc_write.log 2>&1 || exit 1
bash tools/pmc_step.sh ${tag} > gpurun_out/${tag}_pmc_step.txt 2>&1
tail -1 gpurun_out/${tag}_bench_plain.log | cut -c1-400
