"""bench.py prints ONE JSON line with the fields the driver reads (run here on the small room_test workload)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    env = dict(os.environ, OC_BENCH_CPU_BUDGET="1.5", OC_BENCH_NUMPY_STACK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "room_test", "--steps", "60",
                          "--warmup", "5"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["unit"] == "steps/s" and d["value"] > 0 and abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-6 * 1e3
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # `roofline` = the kernel class with the largest share of the timed region's GPU time: an HBM-bound HJB kernel, or the
    # step's force kernel (bound by vector-instruction issue: SURVEY 8d's fp32-vector fraction, HBM fraction beside it --
    # then `roofline_hbm` carries the HBM-bound kernel that matters most)
    r = d["roofline"]
    assert d["dominant_kernel_class"] in ("hjb_stage", "hjb_out", "hjb_list", "gcfm_forces", "gcfm_finish", "density")
    assert r["bound"] in ("hbm", "valu") and "traffic" in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    rh = r if r["bound"] == "hbm" else d["roofline_hbm"]
    assert rh["bound"] == "hbm" and rh["unit"] == "GB/s" and rh["peak"] == 8000.0
    assert abs(rh["frac"] - rh["achieved"] / rh["peak"]) < 1e-12 and 0 < rh["frac"] < 1
    assert d["config"]["hjb_field"].startswith("ring")
    assert "roofline_valu" not in d      # priced from the committed PMC passes of the default workload only
    assert "room_test" in d["metric"] and d["config"]["mode"] == "single"
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "steps/s" and c["sample"]
    assert c["host_cpus"] >= c["cores"] and c["rk_attempts"] > 100
    b = d["breakdown"]
    assert b["kernels"]["hjb_stage"]["achieved_GBps"] < 8000.0          # priced with the bytes the launch pair moves
    assert b["gcfm_kernel_ms_per_step"] > 0


def _launcherless_bench(workload, extra=()):
    """`python bench.py --gpus 2 ...` WITHOUT a launcher (the shape of the driver's single-GPU command with N = 2): bench.py
    starts torch.distributed.run itself as a child process and relays rank 0's line"""
    env = dict(os.environ, OC_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--workload", workload, "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    if out.returncode != 0:                 # (the whole stderr of every rank, for the record: the assertion shows its tail)
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        open(os.path.join(REPO, "gpurun_out", "bench_contract_failure_%s.log" % workload), "w").write(out.stderr)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_two_ranks_without_a_launcher():
    d = _launcherless_bench("exit_opposite", ("--mode", "sharded", "--no-ensemble-line"))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["mode"] == "sharded" and d["value"] > 0
    assert d["rccl"]["ranks"] == 2 and d["rccl"]["hjb_key_plan"] == [[0], [1]]


def _torchrun_bench(workload, extra=(), env_extra=None, nproc=2):
    """the driver's N > 1 command (one rank per GPU) rehearsed on this one-GPU box: both ranks on cuda:0, collectives
    over gloo (OC_BENCH_ONE_DEVICE=1) -- the same bench.py code path, communicators and library calls as over RCCL"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OC_BENCH_ONE_DEVICE="1", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", str(nproc), "--steps", "20", "--warmup", "5",
           "--workload", workload, "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    if out.returncode != 0:                 # (the whole stderr of every rank, for the record: the assertion shows its tail)
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        open(os.path.join(REPO, "gpurun_out", "bench_contract_failure_%s.log" % workload), "w").write(out.stderr)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_two_ranks_sharded_keys_first():
    """2 keys on 2 ranks: each rank solves one key whole (no solve-time collective), agents replicated, one all-reduce
    per step; the line says what RCCL would have carried, and carries the ensemble figure beside it"""
    d = _torchrun_bench("exit_opposite", ("--mode", "sharded"))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["mode"] == "sharded" and d["value"] > 0
    r = d["rccl"]
    # keys on different ranks: the forces by the all-gather of tile shares (5000 agents -> 79 tiles, 40 per rank, 32 B per
    # slot + the block tails), the desired velocities by their own sum (16 B per slot)
    assert r["ranks"] == 2 and r["hjb_key_plan"] == [[0], [1]] and r["collectives_per_step"] == 2
    assert r["step_allgathers"] == 25 and r["step_allgather_bytes"] == 8 * 2 * (40 * 256 + 8)
    assert r["step_allreduces"] == 25 and r["step_allreduce_bytes"] == 16 * 5056
    assert r["hjb_allgathers_this_rank"] == 0 and r["hjb_halo_exchanges_this_rank"] == 0
    assert d["breakdown"]["hjb_solves_this_rank"] == 1
    e = d["ensemble"]
    assert e["scaling"] == "weak" and e["value"] > 0


def test_bench_four_ranks_keys_first_with_the_rows_variant_beside_it():
    """BASELINE configs[3]'s shape -- 2 keys on 4 ranks -- on a small room (4 processes on this one GPU): the line is the
    keys-first plan (each key whole on the first rank of its group), and `rows_variant` is SURVEY 8e's "2 keys x 2-way row
    split" measured on the same ranks: its plan, its halo message, its collectives per attempt."""
    d = _torchrun_bench("exit_opposite", ("--mode", "sharded", "--no-ensemble-line"), nproc=4)
    assert d["n_gpus"] == 4 and d["rccl"]["hjb_key_plan"] == [[0], [2]] and d["value"] > 0
    v = d["rows_variant"]
    assert v["hjb_key_plan"] == [[0, 1], [2, 3]] and v["value"] > 0
    assert v["hjb_halo_message_bytes"] == 2 * 8 * 512 * 4 and v["hjb_halo_exchanges_this_rank"] >= 100
    assert v["hjb_collectives_per_attempt"] == {"allgather_16B_per_rank": 1, "neighbour_halo_exchange": 1}


def test_bench_two_ranks_sharded_row_blocks():
    """1 key on 2 ranks: two row blocks, an all-gather of the error norm and a halo exchange per RK attempt
    (OC_ROW_MIN_CELLS=1: by default a 1024 x 1024 grid is not worth splitting -- parallel.key_row_plan -- and stays whole,
    which the first call checks)"""
    whole = _torchrun_bench("slalom", ("--mode", "sharded", "--no-ensemble-line"))
    assert whole["rccl"]["hjb_key_plan"] == [[0]] and whole["rccl"]["hjb_halo_exchanges_this_rank"] == 0
    # (a single unsplit key is solved whole by EVERY rank: the floor field is local everywhere, one all-gather per step)
    assert whole["rccl"]["collectives_per_step"] == 1 and whole["rccl"]["step_allreduces"] == 0 and whole["value"] > 0
    d = _torchrun_bench("slalom", ("--mode", "sharded", "--no-ensemble-line"), {"OC_ROW_MIN_CELLS": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "ensemble" not in d
    r = d["rccl"]
    assert r["hjb_key_plan"] == [[0, 1]] and r["step_allreduces"] == 25 and r["step_allgathers"] == 25
    assert r["hjb_halo_exchanges_this_rank"] >= 2 * d["breakdown"]["rk_attempts"]                 # warm-up + timed solve
    assert r["hjb_allgathers_this_rank"] == r["hjb_halo_exchanges_this_rank"] + 6
    assert r["hjb_halo_message_bytes"] == 2 * 8 * 1024 * 4
