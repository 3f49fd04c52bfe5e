# needs the diagnostic build: make -C optimal_crowds_amd/csrc diag   (the release library rejects the "diag" knob)
export OC_HIP_LIB=$PWD/optimal_crowds_amd/liboc_hip_diag.so
# where k_forces' instructions go: VALU / LDS counts and time with parts of the kernel cut out (results are wrong in these runs)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for d in 0 1 2 3 5 7; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-include-regex "k_forces" --output-format csv -d gpurun_out/diagf_$d -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-work-count --tune diag=$d > gpurun_out/diagf_$d.log 2>&1
  python - $d <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/diagf_%s/**/*counter_collection.csv' % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("diag=%s " % sys.argv[1] + "  ".join("%s=%.3g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done
