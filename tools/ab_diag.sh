# needs the diagnostic build (make -C optimal_crowds_amd/csrc diag): time of k_forces with parts cut out (results WRONG)
# bits: 1 = no phase 2, 2 = no prep work (wall search), 4 = no compaction either, 8 = no lookups
export OC_HIP_LIB=$PWD/optimal_crowds_amd/liboc_hip_diag.so
for d in ${DIAGS:-0 8 2 10 1 5 7 15}; do STEPS=${STEPS:-200} bash tools/ab_tune.sh "diag=$d"; done
