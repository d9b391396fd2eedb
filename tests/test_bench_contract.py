"""bench.py prints ONE JSON line with the contract's keys; the N>1 code path is exercised with a single rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"]


def _run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--nbath", "9", "--steps", "3", "--warmup", "1", "--no-cpu"] + extra
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout.decode()      # ONE line on stdout, nothing else (library banners go to stderr)
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--storage", "kron", "--no-extra"], ["--force-dist", "--no-extra"], ["--force-dist", "--comm", "torch", "--no-extra", "--storage", "direct"]],
                         ids=["default", "kron", "dist_rccl", "dist_torch"])
def test_bench_line(extra):
    d = _run(extra)
    for k in KEYS:
        assert k in d, k
    assert d["unit"] == "GB/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["time_split_ms"]["hxv_product"] > 0
    if "--force-dist" in extra:      # the exchange step is timed on the comm stream
        assert d["time_split_ms"]["exchange_on_comm_stream_overlapped_with_local_blocks"] >= 0.0
    if not extra:
        assert set(d["other_formats"]) == {"kron", "direct"}
        assert d["ground_state"]["nlanc"] > 10 and d["ground_state"]["seconds"] > 0
