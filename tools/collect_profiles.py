#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the tracked summaries under profiles/ (round tag as argv[1]).

Expects (produced by the commands in profiles/README.md):
  gpurun_out/<tag>_stats/   rocprofv3 --kernel-trace --stats of `python bench.py`
  gpurun_out/<tag>_pmc_fetch/, <tag>_pmc_write/   separate --pmc FETCH_SIZE / WRITE_SIZE passes
  gpurun_out/<tag>_bench.log  stdout of the profiled bench run;  gpurun_out/<tag>_bench_plain.log  un-profiled run
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
workload = sys.argv[2] if len(sys.argv) > 2 else "metro_station"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                           # noqa: E402  (csrc_hash: the stamp bench.py checks)
G = os.path.join(REPO, "gpurun_out")
P = os.path.join(REPO, "profiles")
os.makedirs(P, exist_ok=True)


def short(name):
    for k in ("k_out", "k_forces", "k_finish", "k_pack", "k_tile_lists", "k_cell_keys", "k_edt", "k_wallmap", "k_reduce_final",
              "k_rhs", "k_lincomb", "k_scale", "k_fill", "k_make_coef", "k_norm_scaled", "k_density", "k_control", "k_list",
              "k_ring_fill", "k_ring_ctl", "k_halo_stage", "k_seq_pairs", "k_seq_step"):
        if k + "<" in name or k + "(" in name:
            return k
    if "k_stageV<" in name:
        return "k_stageV(all six)" if ", 6, " in name else ("k_stageV(3,B)" if "true" in name else "k_stageV(3,A)")
    for k in ("k_stage3<", "k_stage6<", "k_stage<"):
        if k in name:
            a = name.index(k)
            return name[a:name.index(">", a) + 1].replace("float, ", "").replace(" ", "")
    return name[:48]


sfx = "" if workload == "metro_station" else "_" + workload
st = sorted(glob.glob(os.path.join(G, tag + sfx + "_stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if st:
    shutil.copy(st[0], os.path.join(P, tag + sfx + "_bench_kernel_stats.csv"))
for name in (tag + sfx + "_bench.log", tag + sfx + "_bench_plain.log", tag + sfx + "_bench_steps20.log"):
    src = os.path.join(G, name)
    if os.path.exists(src):
        lines = [l for l in open(src) if l.startswith("{")]
        if lines:
            open(os.path.join(P, name.replace(".log", ".json")), "w").write(lines[-1])

# the round's other bench lines (tools/run_extras.sh): every gpurun_out/<tag>_<what>.log whose last line is a bench line
if workload == "metro_station":
    for src in sorted(glob.glob(os.path.join(G, tag + "_*.log"))):
        name = os.path.basename(src)
        what = name[len(tag) + 1:-4]
        if not what.startswith(("bench_mixed", "bench_slices", "sequential_", "synchronous_", "rccl_", "rehearsal_")):
            continue
        lines = [l for l in open(src) if l.startswith("{")]
        if lines:
            open(os.path.join(P, name.replace(".log", ".json")), "w").write(lines[-1])

traffic = {}
for d, cname in ((tag + sfx + "_pmc_fetch", "FETCH_SIZE"), (tag + sfx + "_pmc_write", "WRITE_SIZE")):
    f = sorted(glob.glob(os.path.join(G, d, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == cname:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    # the HJB attempt sequence is enqueued ahead of the device-side controller: launches of k_out for rejected attempts
    # and of all three kernels past the end of a solve return at once -- they are not "launches" of the kernel's work
    # (k_out of an accepted step writes > 100 MB, an early exit a few KB)
    def real(k, v):
        if k in ("k_out", "k_stageV(all six)", "k_stageV(3,A)", "k_stageV(3,B)"):
            big = [x for x in v if x > 0.05 * max(v)]
            return big or v
        return v
    traffic[cname] = {k: {"launches": len(real(k, v)), "avg_KB_per_launch": sum(real(k, v)) / len(real(k, v)),
                          "launches_incl_early_exits": len(v), "sum_KB": sum(v),
                          "working_launches": sum(1 for x in v if x > 64.0)}
                      for k, v in sorted(agg.items()) if k.startswith("k_")}
if traffic:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a
    # wide (16 B/lane) coalesced read -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
    for kname, key in (("k_out", "k_out"), ("k_stageV(all six)", "k_stageV"), ("k_forces", "k_forces"), ("k_density", "k_density")):
        kf_ = traffic.get("FETCH_SIZE", {}).get(kname)
        kw_ = traffic.get("WRITE_SIZE", {}).get(kname)
        if kf_ and kw_:
            traffic[key + "_hbm_bytes_per_launch"] = 2 * kf_["avg_KB_per_launch"] * 1024 + kw_["avg_KB_per_launch"] * 1024
    # k_stageV with the active set: a few launches of a solve look at every tile (the first attempt, and two attempts around
    # every change of scale) and carry most of the bytes; the roofline of the sparse launches is quoted per ATTEMPT -- all
    # bytes of all launches over the launches that did any work (an attempt enqueued past a solve's end moves < 64 KB)
    kf_ = traffic.get("FETCH_SIZE", {}).get("k_stageV(all six)")
    kw_ = traffic.get("WRITE_SIZE", {}).get("k_stageV(all six)")
    if kf_ and kw_ and "sum_KB" in kf_ and "sum_KB" in kw_:
        n_att = max(kf_["working_launches"], 1)
        traffic["k_stageV_hbm_bytes_per_attempt"] = (2 * kf_["sum_KB"] + kw_["sum_KB"]) * 1024 / n_att
        traffic["k_stageV_attempts"] = n_att
        traffic["k_stageV_every_tile_launches"] = kf_["launches"]
    traffic["note"] = ("<kernel>_hbm_bytes_per_launch = 2 x FETCH_SIZE (gfx950 correction for 16 B/lane reads) + WRITE_SIZE, "
                       "KB -> bytes, average over the launches of a --steps 20 run (separate --pmc passes; k_stageV: over the launches that "
                       "look at every tile, k_stageV_hbm_bytes_per_attempt: all bytes over all working launches); k_forces reads "
                       "mostly 4-16 B per lane through scattered gathers, so for it the doubled figure is an upper bound")
    traffic["csrc_sha"] = bench.csrc_hash()         # bench.py quotes the file only for the kernels it was taken with
    traffic["workload"] = workload
    json.dump(traffic, open(os.path.join(P, tag + "_pmc_traffic" + ("" if workload == "metro_station" else "_" + workload)
                                         + ".json"), "w"), indent=1)
# SQ counters of the step kernels (tools/pmc_step.sh): mean per launch -> JSON with the same stamp
# (one file per workload, like the traffic passes: the stamp is the workload the counters were TAKEN on)
hjb_txt = os.path.join(G, tag + sfx + "_pmc_hjb.txt")
if os.path.exists(hjb_txt):
    shutil.copy(hjb_txt, os.path.join(P, tag + sfx + "_pmc_hjb_kernels.txt"))
step_txt = os.path.join(G, tag + sfx + "_pmc_step.txt")
if os.path.exists(step_txt):
    import re
    txt = open(step_txt).read()
    shutil.copy(step_txt, os.path.join(P, tag + sfx + "_pmc_step_kernels.txt"))
    step = {"csrc_sha": bench.csrc_hash(), "workload": workload}
    for kn in ("k_forces", "k_finish"):
        if kn in txt:
            blk = txt[txt.index(kn):]
            nxt = [blk.index(o) for o in ("k_forces", "k_finish") if o != kn and o in blk[1:]]
            blk = blk[:min(nxt)] if nxt else blk
            step[kn] = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)\s+n=\d+\s+mean=([0-9.e+]+)", blk)}
    json.dump(step, open(os.path.join(P, tag + sfx + "_pmc_step.json"), "w"), indent=1)
print("profiles:", sorted(os.listdir(P)))

# bench lines of all BASELINE workloads (tools/run_workloads.sh): one compact JSON
wl = {}
for f in sorted(glob.glob(os.path.join(G, tag + "_wl_*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception:
        continue
    name = os.path.basename(f)[len(tag) + 4:-5]
    b, cb = d["breakdown"], d.get("cpu_baseline") or {}
    wl[name] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"],
                "config": d["config"], "roofline_frac": (d["roofline"] or {}).get("frac"),
                "hjb_solve_s_mean": b["hjb_solve_s_total_this_rank"] / max(b["hjb_solves_this_rank"], 1),
                "hjb_solves": b["hjb_solves_this_rank"], "rk_attempts": b["rk_attempts"],
                "kernel_avg_ms": {k: v["avg_ms"] for k, v in b["kernels"].items()},
                "cpu_baseline": {k: cb.get(k) for k in ("value", "cores", "kind", "t_hjb_solve_s", "t_step_s", "host_cpus",
                                                         "cgroup_quota")} if cb else None,
                "numpy_stack": cb.get("numpy_stack") if cb else None}
if wl:
    json.dump(wl, open(os.path.join(P, tag + "_workloads.json"), "w"), indent=1)
