# PMC passes over k_forces|k_finish (run on the GPU box from the repo root): bash tools/pmc_step.sh <tag> [bench flags]
# KERNELS="k_stageV|k_out|k_control" bash tools/pmc_step.sh <tag> ...  collects the same counters for other kernels
tag=$1; shift
KERNELS=${KERNELS:-"k_forces|k_finish"}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1 || true
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-include-regex "$KERNELS" --output-format csv -d $out/p$i -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dense-line "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        kn = next((k for k in ('k_forces', 'k_finish', 'k_stageV', 'k_stage3', 'k_stage6', 'k_out', 'k_control', 'k_list', 'k_density') if k in n), None)
        if kn in ('k_stageV', 'k_stage3'):
            kn += ' B' if 'true' in n else ' A'
        if kn == 'k_stageV A' and 'Li6E' in n or (kn and 'k_stageV' in kn and ', 6,' in n): kn = 'k_stageV all6' 
        if kn: acc[kn][r['Counter_Name']].append(float(r['Counter_Value']))
for kn in acc:
    print(kn)
    for c, v in sorted(acc[kn].items()):
        print('   %-28s n=%d mean=%.4g' % (c, len(v), sum(v) / len(v)))
PY
