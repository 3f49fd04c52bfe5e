# per-kernel mean durations of the HJB launches of a --steps 20 bench run (kernel trace): bash tools/hjb_kernel_times.sh [workload] [tune]
wl=${1:-metro_station}
tune=${2:-hjb_active=1}
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/kt_${wl}_${tune} -o t -- python $root/bench.py --workload $wl --steps 20 --warmup 2 --no-cpu-baseline --no-work-count --no-dense-line --tune $tune > $root/gpurun_out/kt_${wl}_${tune}.log 2>&1 || { echo failed; tail -3 $root/gpurun_out/kt_${wl}_${tune}.log; }
python - $root/gpurun_out/kt_${wl}_${tune} <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    k = next((k for k in ("k_stageV", "k_out", "k_control") if k in n), None)
    if k:
        acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    big = [x for x in v if x > 4.5]
    print("%-10s launches %5d  mean %.1f us  | above 4.5 us: %5d mean %.1f us, total %.2f ms" % (k, len(v), sum(v) / len(v), len(big), sum(big) / max(len(big), 1), sum(v) / 1e3))
# gaps: time between consecutive HJB kernels on the stream
hj = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if any(k in r["Kernel_Name"] for k in ("k_stageV", "k_out", "k_control")))
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(hj, hj[1:]) if 0 <= b[0] - a[1] < 50000]
print("gaps between consecutive HJB kernels: mean %.2f us, total %.2f ms over %d" % (sum(gaps) / len(gaps), sum(gaps) / 1e3, len(gaps)))
PY
