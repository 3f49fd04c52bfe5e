"""Diagnostic build helper: instruments k_pair with s_memtime stamps per phase (stage wait / phase 1 / compaction /
phase 2) and prints per-wavefront averages from oc_gcfm_step when OC_PAIR_STAMPS is set.  Usage:
    python tools/pair_stamps.py apply   # patches optimal_crowds_amd/csrc/gcfm.hip in place (keep a copy!)
Never commit or time the instrumented build (DVFS / fences differ); read the shares only."""
import sys
p = '/root/repo/optimal_crowds_amd/csrc/gcfm.hip'
s = open(p).read()
def rep(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, 1)
rep("    if (it >= S.ntiles) return;                     // wavefronts are independent: no barrier below",
    "    if (it >= S.ntiles) return;\n    unsigned long long T0 = __builtin_amdgcn_s_memtime(), tS = 0, t1 = 0, tC = 0, t2 = 0, tprev; int ntile = 0, npairs = 0;")
rep("        for (;;) {\n            // ---- stage tile jt", "        for (;;) {\n            tprev = __builtin_amdgcn_s_memtime(); ++ntile;\n            // ---- stage tile jt")
STAMP = '            { asm volatile("s_waitcnt vmcnt(%s) lgkmcnt(0)" ::: "memory"); unsigned long long tn = __builtin_amdgcn_s_memtime(); %s += tn - tprev; tprev = tn; }\n'
rep("            // ---- phase 1:", STAMP % ("0", "tS") + "            // ---- phase 1:")
rep("            // ---- compact, lane-major", STAMP % ("63", "t1") + "            // ---- compact, lane-major")
rep("            // ---- phase 2: every lane takes", STAMP % ("63", "tC") + "            npairs += total;\n            // ---- phase 2: every lane takes")
rep("            buf ^= 1;\n            if (!more) break;", STAMP % ("63", "t2") + "            buf ^= 1;\n            if (!more) break;")
rep("    S.f_pair[k] = make_double2(l_accx[w][lane], l_accy[w][lane]);", "    if (lane == 0) { double* dbg = (double*)S.des + (size_t)it * 8; dbg[0] = (double)(__builtin_amdgcn_s_memtime() - T0); dbg[1] = (double)tS; dbg[2] = (double)t1; dbg[3] = (double)tC; dbg[4] = (double)t2; dbg[5] = ntile; dbg[6] = npairs; }\n    S.f_pair[k] = make_double2(l_accx[w][lane], l_accy[w][lane]);")
rep('    oc_timer_end(h, "gcfm_pair", s);\n    oc_timer_begin(h, "gcfm_wall", s);', '''    oc_timer_end(h, "gcfm_pair", s);
    if (getenv("OC_PAIR_STAMPS") && (simu_step % 50) == 10) {
        std::vector<double> hb((size_t)S.ntiles * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hb.data(), S.des, hb.size() * 8, hipMemcpyDeviceToHost);
        double sum[7] = {0}, mx[7] = {0};
        for (int t = 0; t < S.ntiles; ++t) for (int q = 0; q < 7; ++q) { sum[q] += hb[t * 8 + q]; mx[q] = std::max(mx[q], hb[t * 8 + q]); }
        fprintf(stderr, "[pair stamps step %d] per wave avg(max): total %.0f (%.0f) stage %.0f (%.0f) ph1 %.0f (%.0f) compact %.0f (%.0f) ph2 %.0f (%.0f) cycles; tiles %.1f (%.0f) pairs %.0f (%.0f)\\n",
                simu_step, sum[0] / S.ntiles, mx[0], sum[1] / S.ntiles, mx[1], sum[2] / S.ntiles, mx[2], sum[3] / S.ntiles, mx[3], sum[4] / S.ntiles, mx[4], sum[5] / S.ntiles, mx[5], sum[6] / S.ntiles, mx[6]);
    }
    oc_timer_begin(h, "gcfm_wall", s);''')
rep("#include <algorithm>\n\nnamespace {", "#include <algorithm>\n#include <cstdlib>\n\nnamespace {")
open(p, 'w').write(s)
