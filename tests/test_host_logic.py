"""Host-side logic that needs no GPU: the reference's import line, the choice of the HJB state type, bench.py's helpers."""
import copy
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_line_resolves_to_this_package():
    import optimal_crowds
    from optimal_crowds import simulations, optimals, pedestrians          # noqa: F401  (reference README.md:31)
    from optimal_crowds_amd import simulations as s2
    assert simulations is s2 and optimal_crowds.optimals is optimals
    assert hasattr(simulations, "simulation") and hasattr(pedestrians, "ped")
    assert os.path.exists(os.path.join(REPO, "optimal_crowds", "config.json"))    # where the reference opens it (cwd-relative)


def test_hjb_state_dtype_follows_the_float32_range():
    from oracle import oc_oracle as oc
    from optimal_crowds_amd.engine import hjb_state_dtype
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    assert [hjb_state_dtype(cfg, T) for T in (4.0, 20.0, 25.0, 26.0, 60.0)] == ["float32"] * 3 + ["mixed"] * 2
    assert hjb_state_dtype(cfg, 20.0, "float64") == "float64" and hjb_state_dtype(cfg, 20.0, "mixed") == "mixed"
    cfg["hjb_params"]["mu"] = 50.0                      # phi grows 10 x slower: float32 reaches 10 x farther
    assert hjb_state_dtype(cfg, 200.0) == "float32" and hjb_state_dtype(cfg, 300.0) == "mixed"


def test_bench_helpers():
    sys.path.insert(0, REPO)
    import bench
    h = bench.csrc_hash()
    assert len(h) == 16 and h == bench.csrc_hash()
    cores, lim = bench.host_cores()
    assert 1 <= cores <= lim["affinity"] and lim["host_cpus"] >= 1
    assert set(bench.SHARDED_BY_DEFAULT) <= set(bench.WORKLOADS)
    # a PMC summary taken with other kernel sources is never quoted
    assert bench.pmc_file("no_such_pattern_*.json") is None


def test_the_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it"""
    import re
    offenders = []
    for pkg in ("optimal_crowds_amd", "optimal_crowds", "examples"):
        for root, _, files in os.walk(os.path.join(REPO, pkg)):
            for f in files:
                if f.endswith(".py"):
                    txt = open(os.path.join(root, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|oc_oracle|liboc_oracle", txt, re.M):
                        offenders.append(os.path.join(root, f))
    assert offenders == []
    bench_src = open(os.path.join(REPO, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import", bench_src)]
    legs = [bench_src.rfind("\ndef ", 0, u) for u in uses]
    names = {bench_src[l + 5:bench_src.index("(", l + 5)] for l in legs}
    assert names <= {"cpu_baseline", "numpy_stack_baseline"}            # the baseline legs only
    for src in ("csrc/oc_api.hip", "csrc/hjb.hip", "csrc/gcfm.hip", "csrc/density.hip", "csrc/raster.hip", "csrc/sort.hip"):
        txt = open(os.path.join(REPO, "optimal_crowds_amd", src)).read()
        assert not re.search(r'#include\s*[<"][^>"]*oracle|liboc_oracle|occ_', txt)      # (comments may cite the oracle)


def test_source_hash_ignores_comments_and_white_space(tmp_path, monkeypatch):
    """bench.csrc_hash stamps the counter files: rewording a comment must not make them stale, changing code must"""
    sys.path.insert(0, REPO)
    import bench
    d = tmp_path / "optimal_crowds_amd" / "csrc"
    d.mkdir(parents=True)
    (d / "a.hip").write_text("// one\nint f(int x) { return x + 1; }   /* two */\n")
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    h0 = bench.csrc_hash()
    (d / "a.hip").write_text("// another comment\n\nint f(int x)   {\n  return x + 1;\n}\n")
    assert bench.csrc_hash() == h0
    (d / "a.hip").write_text("int f(int x) { return x + 2; }\n")
    assert bench.csrc_hash() != h0


def test_agent_state_dictionary_synchronises_before_a_read():
    """Engine.agents with lazy_state: reading a state array goes through flush_state first; status / key / N do not"""
    from optimal_crowds_amd.engine import _AgentState

    class Eng:
        n = 0

        def flush_state(self):
            self.n += 1
    e = Eng()
    a = _AgentState(e, N=3, x="X", y="Y", vx="VX", vy="VY", agent_time="T", status="S", key="K", v_des="D")
    assert (a["N"], a["status"], a["key"], a["v_des"]) == (3, "S", "K", "D") and e.n == 0
    assert a["x"] == "X" and e.n == 1
    assert [a[k] for k in ("y", "vx", "vy", "agent_time")] == ["Y", "VX", "VY", "T"] and e.n == 5
    assert a.get("x") == "X" and a.get("nope", 7) == 7 and e.n == 6
    assert len(list(a.items())) == 9 and e.n == 7
    assert isinstance(a, dict) and "x" in a                 # (membership tests do not synchronise)
    assert e.n == 7


def test_pmc_traffic_is_quoted_per_attempt_for_the_sparse_stage_launches():
    """bench.pmc_traffic: with the active set the stage launch's roofline quotes all HBM bytes of the solves' stage launches
    over the launches that did any work -- not the mean of the few launches that look at every tile -- and only from a
    file taken on the same workload with the kernel sources in the tree (tools/collect_profiles.py stamps both)."""
    import bench
    d = bench.pmc_file("r*_pmc_traffic*.json", "metro_station")
    if d is None:
        import pytest
        pytest.skip("no PMC summary for the kernel sources in the tree (re-take with tools/run_profiles.sh)")
    assert d["csrc_sha"] == bench.csrc_hash() and d["workload"] == "metro_station"
    per_attempt = bench.pmc_traffic("metro_station", "hjb_stage", per_attempt=True)
    every_tile = bench.pmc_traffic("metro_station", "hjb_stage")
    assert per_attempt == d["k_stageV_hbm_bytes_per_attempt"] and every_tile == d["k_stageV_hbm_bytes_per_launch"]
    assert 0 < per_attempt < every_tile
    assert d["k_stageV_every_tile_launches"] < d["k_stageV_attempts"]
    assert bench.pmc_traffic("slalom", "hjb_stage", per_attempt=True) is None        # another workload: nothing to quote
