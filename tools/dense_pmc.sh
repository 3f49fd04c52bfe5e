# SQ counters of the dense k_stageV launch (GPU box, from the repo root): bash tools/dense_pmc.sh [workload] [knob=value ...]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/pmc_dense
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "k_stageV" --output-format csv -d $out/p$i -- python3 tools/dense_pmc.py "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python3 tools/dense_pmc.py --read $out
