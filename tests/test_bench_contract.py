"""bench.py prints ONE JSON line with the contract's keys; the N>1 code path is exercised with a single rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"]


def _run(extra, env=None, expect_fail=False):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--nbath", "9", "--steps", "3", "--warmup", "1", "--no-cpu"] + extra
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT, env=e)
    if expect_fail:
        assert out.returncode != 0 and not out.stdout.strip(), (out.returncode, out.stdout.decode())
        return out.stderr.decode()
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
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["n_ranks_seen"] == 1 and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "workload" in d["config"] and "model" not in d["config"] and "storage=" in d["config"]["forms"]
    assert d["time_split_ms"]["hxv_product"] > 0
    if "--force-dist" in extra:      # the exchange step is timed on the comm stream
        assert d["time_split_ms"]["exchange_on_comm_stream_overlapped_with_local_blocks"] >= 0.0
    if not extra:
        assert set(d["other_formats"]) == {"kron", "direct"}
        assert d["ground_state"]["nlanc"] > 10 and d["ground_state"]["seconds"] > 0
        # the reference's default solver path (sp_eigh, two pairs) timed next to the plain ground state, same E0
        assert d["eigsh_c3_s"] > 0 and abs(d["default_solver_path"]["E"][0] - d["ground_state"]["E0"]) < 1e-8
        assert "bounded" in r["bound_note"]
    if "--gpus" in extra or "--force-dist" in extra:
        assert "exchange_model_ms" in d and d["exchange_model_ms"]["busiest_pair_at_77GBps"] >= 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("storage", ["csr", "direct"])
def test_bench_emulated_rank_line(storage):
    """`bench.py --emulate-rank r --of P`: one rank of P alone on the GPU (measurement aid): one JSON line that says so, with the rank's
    share of the rows, its exchange volume and the time split; never a job throughput (n_gpus stays 1, no `scaling` claim)."""
    d = _run(["--emulate-rank", "2", "--of", "4", "--storage", storage])
    assert d["emulated"] == {"rank": 2, "of": 4, "exchange": d["emulated"]["exchange"]} and "EMULATED" in d["metric"]
    assert d["n_gpus"] == 1 and "scaling" not in d and d["ms_per_step"] > 0
    assert d["config"]["rows_of_rank"] * 4 <= d["config"]["N"] + 4 * 252 and d["config"]["rows_of_rank"] > 0
    assert d["exchange"]["mode"] == "needed-rows" and d["exchange"]["received_MB_per_product"] > 0
    assert d["time_split_ms"]["hxv_product_of_rank"] > 0 and d["time_split_ms"]["emulated_exchange_on_comm_stream"] >= 0


@pytest.mark.gpu
@pytest.mark.parametrize("storage", ["csr", "kron"])
def test_bench_starts_its_own_ranks(storage):
    """`python bench.py --gpus 2` with NO launcher around it: two rank processes come up (sharing the one GPU of the test
    box through the host-staged exchange), rank 0 prints the line, and the line counts two ranks -- through an all-reduce
    over the communicator the products use."""
    d = _run(["--gpus", "2", "--share-gpus", "--no-extra", "--storage", storage])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "strong"
    assert "x2" in d["config"]["parallelism"] and "needed-rows" in d["config"]["parallelism"]
    x = d["exchange_bytes_per_product"]
    assert x["mode"] == "needed-rows" and len(x["send_matrix_MB"]) == 2 and x["send_matrix_MB"][0][0] == 0 and x["send_matrix_MB"][0][1] > 0
    r = d["strong_scaling_reference"]
    assert r["ms_per_step_1gpu"] > 0 and r["speedup_vs_1gpu"] > 0
    assert d["time_split_ms"]["exchange_on_comm_stream_overlapped_with_local_blocks"] >= 0.0


@pytest.mark.gpu
def test_bench_refuses_to_report_fewer_gpus_than_asked():
    # counted in a child: importing torch HERE, after other tests have loaded libedgpu.so against the system HIP / RCCL, would
    # put a second copy of both runtimes into this process (PyTorch bundles its own) and their exit handlers collide
    ndev = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE, timeout=300).stdout.split()[-1])
    if ndev >= 2:
        pytest.skip("needs a box with fewer devices than ranks")
    err = _run(["--gpus", "2", "--no-extra"], expect_fail=True)
    assert "only 1 device" in err
    err = _run(["--gpus", "2", "--no-extra"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, expect_fail=True)
    assert "refusing to report a different GPU count" in err


@pytest.mark.gpu
def test_wrong_import_order_exits_cleanly():
    """edgpu first (a whole ground-state search), THEN `import torch` and a CUDA call: the order that ended round 2's full run
    in "double free or corruption" at interpreter exit.  One HIP runtime is mapped, results are right, exit code 0."""
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, edgpu\n"
        "from edgpu import model, capi\n"
        "m, nu, nd = model.half_filled_c(5)\n"
        "G = edgpu.HvSector(m, nu, nd); e0, x, n = G.sp_lanc_eigh(nitermax=200, threshold=1e-14); G.close()\n"
        "import torch\n"
        "t = torch.ones(1024, device='cuda').sum().item()\n"
        "G = edgpu.HvSector(m, nu, nd, storage='csr'); e1, x, n = G.sp_lanc_eigh(nitermax=200, threshold=1e-14)\n"
        "print('E0 %%.12f %%.12f SUM %%d RUNTIMES %%d' %% (e0, e1, t, len(capi.mapped_hip_runtimes())))\n"
    ) % (ROOT, os.path.join(ROOT, "dmft-lanc-ed_amd"))
    env = dict(os.environ); env.pop("EDGPU_HIP_RUNTIME", None)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out
    assert "SUM 1024 RUNTIMES 1" in out, out
    e0, e1 = [float(x) for x in out.split("E0")[1].split()[:2]]
    assert abs(e0 - (-7.39968598202036)) < 1e-10 and abs(e1 - e0) < 1e-10       # SURVEY 8c known answer of config 1


def test_bench_launcher_needs_no_gpu_to_refuse():
    """CPU: with no device visible the launcher path exits non-zero without a line (it never reports n_gpus < --gpus)."""
    err = _run(["--gpus", "2", "--no-extra"], expect_fail=True)
    assert "GPU" in err or "device" in err


def test_traffic_stamp_covers_the_product_kernel_sources():
    """CPU: the digest that ties roofline.traffic to the committed counter summary is taken over the files that define the product
    kernels, their launches and the storage forms -- every one of them exists, the solver files are not among them (a change to a
    solver loop must not null a traffic figure that is still that of the running kernels), scripts/pmc_traffic.py stamps with the
    same digest, and the committed summary carries one record per storage with the fields bench.py reads."""
    sys.path.insert(0, ROOT)
    import bench
    import importlib.util
    csrc = os.path.join(ROOT, "dmft-lanc-ed_amd", "csrc")
    for f in bench.PRODUCT_SOURCES:
        assert os.path.isfile(os.path.join(csrc, f)), f
    assert "kernels.hpp" in bench.PRODUCT_SOURCES and "product.hpp" in bench.PRODUCT_SOURCES
    assert not {"lanczos.hpp", "basis_kernels.hpp", "tridiag_host.hpp", "edgpu.hip"} & set(bench.PRODUCT_SOURCES)
    src = open(os.path.join(ROOT, "scripts", "pmc_traffic.py")).read()
    assert repr(tuple(bench.PRODUCT_SOURCES)).replace("'", '"') in src.replace("'", '"')
    # no product kernel lives in the solver headers
    for f in ("basis_kernels.hpp", "lanczos.hpp"):
        text = open(os.path.join(csrc, f)).read()
        for k in ("k_csr_il", "k_dwpass2", "k_uppass2", "k_uppass_halo"):
            assert "void %s" % k not in text and "__launch_bounds__(64) %s" % k not in text, (f, k)
    recs = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert sorted(r["storage"] for r in recs) == ["csr", "direct", "kron"]
    for r in recs:
        assert r["hbm_bytes_per_launch"] > 0 and len(r["source_stamp"]) == 16 and r["kernels"]
    assert len(bench.source_stamp()) == 16
