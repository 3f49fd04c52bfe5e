"""bench.py prints ONE JSON line with the fields the driver reads (run here on the small room_test workload)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    env = dict(os.environ, OC_BENCH_CPU_BUDGET="1.5")
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
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    assert "roofline_valu" not in d      # priced from the committed PMC passes of the default workload only
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "steps/s" and c["sample"]
