"""Diagnostic build helper (never commit or time the instrumented build): stamps every wavefront of k_forces with
s_memrealtime (100 MHz) at its start and end and prints the distribution from oc_gcfm_step when OC_WAVE_STAMPS is
set.  Usage:  python tools/wave_stamps.py   (patches optimal_crowds_amd/csrc/gcfm.hip in place; `git checkout` it after),
make -C optimal_crowds_amd/csrc, then on the GPU box  OC_WAVE_STAMPS=1 python bench.py --steps 100 --no-cpu-baseline."""
p = 'optimal_crowds_amd/csrc/gcfm.hip'
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, old[:60]
    s = s.replace(old, new)


rep("    const Snap& C = S.snap;\n    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;\n    if ((int)blockIdx.x < nprep) {",
    "    const Snap& C = S.snap;\n    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;\n"
    "    const unsigned long long T0 = __builtin_amdgcn_s_memrealtime();\n"
    "    long long* stamp = S.widx + 2 * ((size_t)blockIdx.x * 4 + w);\n    if ((int)blockIdx.x < nprep) {")
rep("                          sh.prep_lookup != 0 && (mine || sh.lookup_all) && !OC_DIAG(P, 8u));\n        }\n        return;",
    "                          sh.prep_lookup != 0 && (mine || sh.lookup_all) && !OC_DIAG(P, 8u));\n        }\n"
    "        if (lane == 0) { stamp[0] = (long long)T0; stamp[1] = (long long)__builtin_amdgcn_s_memrealtime(); }\n        return;")
rep("        xs(S, S.des, S.ts_des, k) = ds;\n    }\n}\n\n#define RECT_TAB",
    "        xs(S, S.des, S.ts_des, k) = ds;\n    }\n"
    "    if (lane == 0) { stamp[0] = (long long)T0; stamp[1] = (long long)__builtin_amdgcn_s_memrealtime(); }\n}\n\n#define RECT_TAB")
rep('    oc_timer_end(h, "gcfm_forces", s);',
    '''    oc_timer_end(h, "gcfm_forces", s);
    if (getenv("OC_WAVE_STAMPS") && (simu_step % 50) == 30) {
        const int nw = (nprep + npair) * 4;
        std::vector<long long> hb((size_t)nw * 2);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hb.data(), S.widx, hb.size() * 8, hipMemcpyDeviceToHost);
        long long tmin = hb[0], tmax = hb[1];
        for (int q = 0; q < nw; ++q) { tmin = std::min(tmin, hb[2 * q]); tmax = std::max(tmax, hb[2 * q + 1]); }
        auto report = [&](const char* name, int a, int b) {
            std::vector<double> st, en, du;
            for (int q = a; q < b; ++q) { st.push_back((hb[2 * q] - tmin) * 0.01); en.push_back((hb[2 * q + 1] - tmin) * 0.01); du.push_back((hb[2 * q + 1] - hb[2 * q]) * 0.01); }
            auto pct = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
            fprintf(stderr, "[stamps step %d %s n=%d] start us p0/50/90/100: %.1f %.1f %.1f %.1f | end: %.1f %.1f %.1f %.1f | dur: %.1f %.1f %.1f %.1f\\n", simu_step, name, b - a,
                    pct(st, 0), pct(st, .5), pct(st, .9), pct(st, 1), pct(en, 0), pct(en, .5), pct(en, .9), pct(en, 1), pct(du, 0), pct(du, .5), pct(du, .9), pct(du, 1));
        };
        report("prep", 0, nprep * 4);
        report("pair", nprep * 4, nw);
        {   // pair blocks by dispatch order (deciles): start, duration, end of their wavefronts and the tiles' measured work
            std::vector<int> ord(S.ntiles), wk(S.ntiles);
            (void)hipMemcpy(ord.data(), S.tile_order, 4 * (size_t)S.ntiles, hipMemcpyDeviceToHost);
            (void)hipMemcpy(wk.data(), S.tile_work, 4 * (size_t)S.ntiles, hipMemcpyDeviceToHost);
            const int tpbk = 4 / ns;
            for (int d = 0; d < 10; ++d) {
                const int b0 = npair * d / 10, b1 = npair * (d + 1) / 10;
                double st = 0, du = 0, en = 0, enmax = 0, w = 0; int n = 0, nw2 = 0;
                for (int b = b0; b < b1; ++b) {
                    for (int q = 0; q < 4; ++q) {
                        const int wv = (nprep + b) * 4 + q;
                        const double a = (hb[2 * wv] - tmin) * 0.01, e = (hb[2 * wv + 1] - tmin) * 0.01;
                        st += a; du += e - a; en += e; enmax = std::max(enmax, e); ++n;
                    }
                    for (int q = 0; q < tpbk; ++q) if (b * tpbk + q < S.ntiles) { w += wk[ord[b * tpbk + q]]; ++nw2; }
                }
                fprintf(stderr, "[stamps pair decile %d] blocks %d-%d start %.1f dur %.1f end %.1f (max %.1f) work/tile %.0f\\n", d, b0, b1, st / n, du / n, en / n, enmax, w / std::max(nw2, 1));
            }
        }
        // resident wavefronts over the launch, in 5 us bins (prep | pair)
        const int nbin = (int)((tmax - tmin) * 0.01 / 5.0) + 1;
        fprintf(stderr, "[stamps resident waves per 5 us bin, prep|pair]");
        for (int bnum = 0; bnum < nbin; ++bnum) {
            const double t = (bnum + 0.5) * 5.0;
            int cp = 0, cq = 0;
            for (int q = 0; q < nw; ++q)
                if ((hb[2 * q] - tmin) * 0.01 <= t && (hb[2 * q + 1] - tmin) * 0.01 > t) (q < nprep * 4 ? cp : cq)++;
            fprintf(stderr, " %d|%d", cp, cq);
        }
        fprintf(stderr, "\\n");
    }''')
rep("#include <algorithm>\n", "#include <algorithm>\n#include <cstdlib>\n#include <vector>\n")
open(p, 'w').write(s)
