"""Where a pair wavefront of k_forces spends its LIFE (diagnostic; never commit or time the instrumented build): copies csrc to
/tmp/csrc_phases, stamps pair_share with s_memrealtime at its phase boundaries, builds optimal_crowds_amd/liboc_hip_alt.so.

    python tools/wave_phases.py && <on the GPU box> OC_HIP_LIB=$PWD/optimal_crowds_amd/liboc_hip_alt.so OC_WAVE_PHASES=1 \
        python bench.py --steps 100 --no-cpu-baseline --no-work-count --no-dense-line

prints, for steps 30 and 80: per pair wavefront the mean microseconds in  wait/fetch before a virtual tile is staged | staging +
phase 1 (masks) | units + search | phase 2 (walk) | rest (block barriers, lookups), and the prep wavefronts' lifetimes."""
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "optimal_crowds_amd", "csrc")
dst = "/tmp/csrc_phases"
shutil.rmtree(dst, ignore_errors=True)
shutil.copytree(src, dst, ignore=shutil.ignore_patterns("*.o"))
p = os.path.join(dst, "gcfm.hip")
s = open(p).read()


def rep(old, new, cnt=1):
    global s
    assert s.count(old) == cnt, (s.count(old), old[:70])
    s = s.replace(old, new)


hdr = open(os.path.join(dst, "oc_internal.h")).read().replace('"../../include/oc_hip.h"', '"%s/include/oc_hip.h"' % REPO)
open(os.path.join(dst, "oc_internal.h"), "w").write(hdr)
rep("    const int j0 = sub * NJ;                        // this share's records of a j-tile: j0 .. j0 + NJ - 1",
    "    const int j0 = sub * NJ;\n"
    "    unsigned long long ph_t = __builtin_amdgcn_s_memrealtime(), ph_a0 = 0, ph_a1 = 0, ph_a2 = 0, ph_a3 = 0;\n"
    "    const unsigned long long ph_t0 = ph_t;\n"
    "#define PH(acc) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); acc += n_ - ph_t; ph_t = n_; } while (0)\n")
rep("            // ---- stage the virtual tile (registers -> wave-private LDS; LDS ops of one wavefront complete in order,",
    "            PH(ph_a0);\n            // ---- stage the virtual tile (registers -> wave-private LDS; LDS ops of one wavefront complete in order,")
rep("            {\n                uint32_t m0 = mw[0], m1 = mw[1];\n                if (OC_DIAG(P, 4u)) m0 = m1 = 0u;",
    "            {\n                PH(ph_a1);\n                uint32_t m0 = mw[0], m1 = mw[1];\n                if (OC_DIAG(P, 4u)) m0 = m1 = 0u;")
rep("                    int il_cur = -1;\n                    double2 pa = make_double2(0.0, 0.0);\n                    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), aux = va;\n                    // the share's sums",
    "                    PH(ph_a2);\n                    int il_cur = -1;\n                    double2 pa = make_double2(0.0, 0.0);\n                    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), aux = va;\n                    // the share's sums")
rep("                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n                }\n            }\n#else",
    "                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n                    PH(ph_a3);\n                }\n            }\n#else")
rep("    if (lane == 0) atomicAdd(&S.tile_work[it], work);\n}",
    "    if (lane == 0) atomicAdd(&S.tile_work[it], work);\n"
    "    if (lane == 0) { long long* st_ = S.widx + 8 * ((size_t)blockIdx.x * 4 + w); if (st_[0] == 0) st_[0] = (long long)ph_t0; st_[1] += (long long)ph_a0; st_[2] += (long long)ph_a1;\n"
    "        st_[3] += (long long)ph_a2; st_[4] += (long long)ph_a3; st_[5] = (long long)__builtin_amdgcn_s_memrealtime(); }\n}")
# kernel start / end stamps (slots 6, 7) for every wavefront
rep("    const Snap& C = S.snap;\n    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;\n    if ((int)blockIdx.x < nprep) {",
    "    const Snap& C = S.snap;\n    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;\n"
    "    long long* stamp = S.widx + 8 * ((size_t)blockIdx.x * 4 + w);\n"
    "    if (lane == 0) { stamp[6] = (long long)__builtin_amdgcn_s_memrealtime(); stamp[0] = stamp[1] = stamp[2] = stamp[3] = stamp[4] = stamp[5] = 0; }\n    if ((int)blockIdx.x < nprep) {")
rep("                          sh.prep_lookup != 0 && (mine || sh.lookup_all) && !OC_DIAG(P, 8u));\n        }\n        return;",
    "                          sh.prep_lookup != 0 && (mine || sh.lookup_all) && !OC_DIAG(P, 8u));\n        }\n"
    "        if (lane == 0) stamp[7] = (long long)__builtin_amdgcn_s_memrealtime();\n        return;")
rep("        xs(S, S.des, S.ts_des, k) = ds;\n    }\n}\n\n__device__ __forceinline__ double readlane_f64",
    "        xs(S, S.des, S.ts_des, k) = ds;\n    }\n    if (lane == 0) stamp[7] = (long long)__builtin_amdgcn_s_memrealtime();\n}\n\n__device__ __forceinline__ double readlane_f64")
rep('    oc_timer_end(h, "gcfm_forces", s);',
    '''    oc_timer_end(h, "gcfm_forces", s);
    if (getenv("OC_WAVE_PHASES") && (simu_step % 50) == 30) {
        const int nw = (nprep + npair) * 4;
        std::vector<long long> hb((size_t)nw * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hb.data(), S.widx, hb.size() * 8, hipMemcpyDeviceToHost);
        long long tmin = hb[6], tmax = hb[7];
        for (int q = 0; q < nw; ++q) { tmin = std::min(tmin, hb[8 * q + 6]); tmax = std::max(tmax, hb[8 * q + 7]); }
        double a[6] = {0, 0, 0, 0, 0, 0}, life = 0, prep_life = 0, prep_max = 0; int np = 0;
        for (int q = nprep * 4; q < nw; ++q) {
            const long long* r = &hb[8 * q];
            if (r[5] == 0) continue;
            a[0] += r[1] * 0.01; a[1] += r[2] * 0.01; a[2] += r[3] * 0.01; a[3] += r[4] * 0.01;
            a[4] += (r[0] - r[6]) * 0.01;           // kernel entry -> pair_share entry (tile load, first barrier)
            a[5] += (r[7] - r[5]) * 0.01;           // pair_share exit -> kernel exit (last barrier, sums out, lookups)
            life += (r[7] - r[6]) * 0.01; ++np;
        }
        for (int q = 0; q < nprep * 4; ++q) { const double d = (hb[8 * q + 7] - hb[8 * q + 6]) * 0.01; prep_life += d; prep_max = std::max(prep_max, d); }
        fprintf(stderr, "[phases step %d] launch %.1f us; %d pair wavefronts, mean life %.1f us: before pair_share %.1f | wait/fetch before a staged tile %.1f | stage + phase 1 %.1f | units + search %.1f | phase 2 %.1f | after %.1f ;  prep wavefronts: mean life %.1f us, max %.1f\\n",
                simu_step, (tmax - tmin) * 0.01, np, life / np, a[4] / np, a[0] / np, a[1] / np, a[2] / np, a[3] / np, a[5] / np, prep_life / std::max(nprep * 4, 1), prep_max);
    }''')
rep("#include <algorithm>\n", "#include <algorithm>\n#include <cstdlib>\n#include <vector>\n")
open(p, "w").write(s)
# carve(): widx must hold 8 words per wavefront
objs = []
for f in sorted(os.listdir(dst)):
    if f.endswith(".hip"):
        o = os.path.join(dst, f[:-4] + ".o")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function",
                               "-c", os.path.join(dst, f), "-o", o])
        objs.append(o)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                       os.path.join(REPO, "optimal_crowds_amd", "liboc_hip_alt.so")] + objs + ["-ldl"])
print("built optimal_crowds_amd/liboc_hip_alt.so with phase stamps")
