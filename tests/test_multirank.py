"""The N>1 path: dw-column partition + exchange.  CPU: world_size-2 gloo job over the host
logic.  GPU: world_size-2 and -3 jobs sharing the one GPU through the callback exchange, and
the RCCL code path on a single-rank communicator."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _launch(mode, case, size, timeout=150, extra_env=None):
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for r in range(size):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.update(extra_env or {})
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "mp_worker.py"), mode, case, d], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        # poll: a rank that dies leaves the others blocked in a collective -- stop them at once and show its output
        import time
        t0 = time.time(); failed = None
        while True:
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = bad[0]; break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > timeout:
                failed = -1; break
            time.sleep(0.2)
        if failed is not None:
            for q in procs:
                if q.poll() is None:
                    q.kill()
        outs = [p.communicate()[0].decode() for p in procs]
        assert failed is None, ("timeout" if failed == -1 else "rank %d failed" % failed) + ":\n" + "\n-----\n".join(o[-2500:] for o in outs)
        return [json.load(open(os.path.join(d, "rank%d.json" % r))) for r in range(size)]


@pytest.mark.parametrize("case,size", [("c_nb5_half", 2), ("phonons_nb3_nph3", 2), ("nb5_offhalf_2_4", 3)])
def test_partition_and_staged_exchange_gloo_cpu(case, size):
    res = _launch("cpu", case, size)
    assert all(r["ok"] for r in res)


@pytest.mark.parametrize("case,size", [("c_nb5_half", 2), ("c_nb7_half", 3), ("nb5_offhalf_2_4", 4)])
def test_needed_rows_exchange_lists_gloo_cpu(case, size):
    """world_size 2-4 on CPU: send/receive row lists derived independently on both sides from H_dw are consistent, and the
    exchange delivers exactly the rows a rank's dw hops read (and fewer than an all-gather would)."""
    res = _launch("cpu_halo", case, size)
    assert all(r["ok"] for r in res)
    assert all(r["halo_rows"] <= r["allgather_rows"] for r in res)


@pytest.mark.parametrize("case,size", [("c_nb5_half", 2), ("nb5_offhalf_2_4", 3)])
def test_transpose_scheme_block_arithmetic_gloo_cpu(case, size):
    """world_size 2 and 3 on CPU: the two all-to-alls of the distributed-transpose dw pass with uneven up/dw splits."""
    res = _launch("cpu_transpose", case, size)
    assert all(r["ok"] for r in res)


@pytest.mark.gpu
@pytest.mark.parametrize("case,size", [("c_nb5_half", 2), ("c_nb7_half", 3), ("norb2_normal_JxJp", 2), ("phonons_norb2_JxJp", 2), ("nb5_offhalf_2_4", 4),
                                       ("orbs_norb2_nb3", 2), ("orbs_norb2_nb3_uneven", 3), ("phonons_orbs", 2),
                                       ("replica_sz_sx_JxJp", 3), ("norb3_nb1_JxJp", 2), ("norb2_hybrid_JxJp", 4)])
def test_multirank_on_one_gpu_callback_exchange(edgpu_mod, case, size):
    import cases
    from oracle import oracle
    res = _launch("gpu", case, size)
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    e_dense = np.linalg.eigvalsh(S.hmat())[0]
    G = edgpu_mod.HvSector(m, nu, nd)
    v0 = np.random.RandomState(5).uniform(0, 1, S.Dim)
    a1, b1, _ = G.sp_lanc_tridiag(v0, 20)
    G.close()
    for r in res:
        for k, v in r.items():
            if k.startswith("e0_"):
                assert abs(v - e_dense) < 1e-10, (k, v, e_dense)
            if k.startswith("a_"):   # same coefficients on every rank and as the single-rank run
                np.testing.assert_allclose(v, a1, rtol=1e-9, atol=1e-9)
            if k.startswith("b_"):
                np.testing.assert_allclose(v, b1, rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("case,size", [("c_nb7_half", 2), ("nb5_offhalf_2_4", 3)])
def test_multirank_large_sector_forms(edgpu_mod, case, size):
    """The storage forms that switch on for large sectors, forced on small ones with several ranks: two-pass stored
    product, and the explicit CSR as three blocks (local | dw hops into own rows | dw hops into received rows)."""
    res = _launch("gpu", case, size, extra_env={"EDGPU_KRON_SPLIT": "1", "EDGPU_CSR_SPLIT": "1"})
    name, m, nu, nd = [z for z in __import__("cases").zoo() if z[0] == case][0]
    from oracle import oracle
    e_dense = np.linalg.eigvalsh(oracle.Sector(m, nu, nd).hmat())[0]
    for r in res:
        for k, v in r.items():
            if k.startswith("e0_"):
                assert abs(v - e_dense) < 1e-10, (k, v, e_dense)


@pytest.mark.gpu
@pytest.mark.parametrize("case,size,env", [("c_nb7_half", 2, {"EDGPU_HALO": "0"}), ("nb5_offhalf_2_4", 3, {"EDGPU_HALO": "0", "EDGPU_KRON_SPLIT": "1", "EDGPU_CSR_SPLIT": "1"}),
                                           ("c_nb7_half", 3, {"EDGPU_KRON_SPLIT": "1", "EDGPU_CSR_SPLIT": "1", "EDGPU_CSR_THREE": "0"}),
                                           ("nb5_offhalf_2_4", 2, {"EDGPU_KRON_SPLIT": "1", "EDGPU_CSR_SPLIT": "1", "EDGPU_CSR_PITCH_MULTI": "0"})])
def test_multirank_fallback_forms(edgpu_mod, case, size, env):
    """The fallbacks of the several-rank forms that are on by default: all-gather of the input vector instead of the needed-rows
    exchange (EDGPU_HALO=0, the reference's allgather_vector_MPI), the explicit CSR of a rank in two blocks + own-row copy
    instead of three (EDGPU_CSR_THREE=0), dense rows in the x-space and in the messages (EDGPU_CSR_PITCH_MULTI=0)."""
    res = _launch("gpu", case, size, extra_env=env)
    name, m, nu, nd = [z for z in __import__("cases").zoo() if z[0] == case][0]
    from oracle import oracle
    e_dense = np.linalg.eigvalsh(oracle.Sector(m, nu, nd).hmat())[0]
    for r in res:
        for k, v in r.items():
            if k.startswith("e0_"):
                assert abs(v - e_dense) < 1e-10, (env, k, v, e_dense)


@pytest.mark.gpu
@pytest.mark.parametrize("case,size,env", [("c_nb7_half", 3, {}), ("nb5_offhalf_2_4", 4, {}), ("nb6_even", 2, {"EDGPU_DEBUG": "4"}),
                                           ("nspin2_random_bath", 3, {}), ("norb2_normal_offdiag_hloc", 2, {}),
                                           ("c_nb7_half", 3, {"EDGPU_TR_CHUNKS": "1"}), ("c_nb7_half", 4, {"EDGPU_TR_CHUNKS": "7"}),
                                           ("nb5_offhalf_2_4", 3, {"EDGPU_TR_CHUNKS": "32"})])
def test_multirank_distributed_transpose(edgpu_mod, case, size, env):
    """EDGPU_DW_TRANSPOSE=1: the dw hops through the reference's own scheme (transpose, all-to-all, local pass along
    rows, transpose back) instead of the needed-rows exchange; uneven up/dw splits, rows with and without LDS staging.
    The exchange is pipelined over chunks of the up-block rows (default 4; EDGPU_TR_CHUNKS=1 is the serial form, 7 and 32 give
    uneven and single-row chunks): every form must reach the dense ground-state energy."""
    res = _launch("gpu", case, size, extra_env=dict({"EDGPU_DW_TRANSPOSE": "1"}, **env))
    name, m, nu, nd = [z for z in __import__("cases").zoo() if z[0] == case][0]
    from oracle import oracle
    e_dense = np.linalg.eigvalsh(oracle.Sector(m, nu, nd).hmat())[0]
    for r in res:
        for k, v in r.items():
            if k.startswith("e0_"):
                assert abs(v - e_dense) < 1e-10, (k, v, e_dense)


@pytest.mark.gpu
@pytest.mark.parametrize("case,size,env", [("c_nb7_half", 2, {}), ("nb5_offhalf_2_4", 3, {"EDGPU_KRON_SPLIT": "1", "EDGPU_CSR_SPLIT": "1"}),
                                           ("norb2_normal_JxJp", 2, {}), ("phonons_orbs", 2, {})])
def test_multirank_one_allreduce_per_iteration(edgpu_mod, case, size, env):
    """EDGPU_ONE_ALLREDUCE=1: the plain Lanczos loops of several ranks carry the norm of the newest vector as a per-rank partial
    sum and add it to the next product's dot in one 16-byte all-reduce (lz_iter_deferred).  Same ground-state energy as the dense
    matrix (the invariant-subspace test looks one product late), same tridiagonal coefficients as a single rank, and about half
    the all-reduce calls of the two-call form for a 20-step tridiagonalisation."""
    import cases
    from oracle import oracle
    res1 = _launch("gpu", case, size, extra_env=dict({"EDGPU_ONE_ALLREDUCE": "1"}, **env))
    res0 = _launch("gpu", case, size, extra_env=dict(env))
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    e_dense = np.linalg.eigvalsh(S.hmat())[0]
    G = edgpu_mod.HvSector(m, nu, nd)
    a1, b1, _ = G.sp_lanc_tridiag(np.random.RandomState(5).uniform(0, 1, S.Dim), 20)
    G.close()
    for r1, r0 in zip(res1, res0):
        for k, v in r1.items():
            if k.startswith("e0_"):
                assert abs(v - e_dense) < 1e-10, (k, v, e_dense)
            if k.startswith("a_"):
                np.testing.assert_allclose(v, a1, rtol=1e-9, atol=1e-9)
            if k.startswith("b_"):
                np.testing.assert_allclose(v, b1, rtol=1e-9, atol=1e-9)
            if k.startswith("allreduce_calls_tridiag20_"):
                n20 = min(20, S.Dim)
                assert r0[k] >= 2 * n20 and v <= r0[k] - n20 // 2 - 2, (k, v, r0[k])      # 1 start + 20 x 2 against 1 start + (7 merged + 1 + 1 flush) per chunk of 8
                print(case, size, k, "all-reduce calls", r0[k], "->", v)


@pytest.mark.gpu
@pytest.mark.parametrize("case,size,env", [("c_nb7_half", 3, {}), ("c_nb7_half", 4, {}), ("nb5_offhalf_2_4", 4, {"EDGPU_KRON_SPLIT": "1", "EDGPU_CSR_SPLIT": "1"}),
                                           ("norb2_normal_JxJp", 3, {}), ("nb6_even", 5, {}),
                                           ("c_nb7_half", 4, {"EDGPU_ONE_ALLREDUCE": "1", "EDGPU_KRON_SPLIT": "1", "EDGPU_CSR_SPLIT": "1"})])      # all opt-in forms together, pitched CSR rows
def test_multirank_relay_routing(edgpu_mod, case, size, env):
    """EDGPU_RELAY=1: the needed-rows exchange in two phases with the heavy pairs handing part of their rows to a third rank
    (relay_plan).  The x-space a rank ends up with is the same, so everything downstream is: dense ground-state energy, the
    single-rank Lanczos coefficients; the plan is the same on every rank, moves every row (those that are relayed over two
    links) and never loads the busiest link more than the plain exchange does."""
    import cases
    from oracle import oracle
    res = _launch("gpu", case, size, extra_env=dict({"EDGPU_RELAY": "1"}, **env))
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    e_dense = np.linalg.eigvalsh(S.hmat())[0]
    G = edgpu_mod.HvSector(m, nu, nd)
    a1, b1, _ = G.sp_lanc_tridiag(np.random.RandomState(5).uniform(0, 1, S.Dim), 20)
    G.close()
    for r in res:
        for k, v in r.items():
            if k.startswith("e0_"):
                assert abs(v - e_dense) < 1e-10, (k, v, e_dense)
            if k.startswith("a_"):
                np.testing.assert_allclose(v, a1, rtol=1e-9, atol=1e-9)
            if k.startswith("b_"):
                np.testing.assert_allclose(v, b1, rtol=1e-9, atol=1e-9)
    for st in ("kron", "direct", "csr"):
        infos = [r["relay_" + st] for r in res]
        assert all(i == infos[0] for i in infos)                       # one plan, derived independently on every rank
        i0 = infos[0]
        assert i0["mode"] == "needed-rows" and i0["enabled"]
        D, R = np.array(i0["direct"]), np.array(i0["routed"])
        assert i0["before"] == D.max() and i0["after"] <= i0["before"]
        assert R.sum() >= D.sum() and (R.sum() > D.sum()) == (i0["chunks"] > 0)
        assert (np.diag(R) == 0).all()
        print(case, size, st, "busiest link", i0["before"], "->", i0["after"], "rows,", i0["chunks"], "chunks")


@pytest.mark.gpu
@pytest.mark.parametrize("case,size", [("c_nb5_half", 2), ("norb2_normal_JxJp", 3), ("orbs_norb2_nb3", 2), ("phonons_nb3_nph3", 2), ("phonons_orbs", 2)])
def test_multirank_legs_observables_twin(edgpu_mod, case, size):
    """GF / chi / phonon legs, impurity histogram + density matrix and the twin-sector hand-off with the
    state split over ranks: same numbers as the single-rank library and the oracle's apply_op."""
    res = _launch("legs", case, size)
    assert all(r["ok"] for r in res) and res[0]["nlegs"] >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("st", ["kron", "direct", "csr"])
def test_emulated_rank_of_p_runs_the_multi_rank_forms_alone(edgpu_mod, monkeypatch, st):
    """edgpu_comm_create_emulated (bench.py --emulate-rank): one rank of P alone on the GPU, exchanges replaced by device copies of the same
    byte count, all-reduces local.  A measurement aid -- its numbers are not a result -- so the test checks what it must guarantee:
    the sector has the partition and exchange lists of the P-rank job (same byte matrix as the callback job of the relay test would
    report for that rank), products and Lanczos steps run to completion on finite numbers, and rank 0 of 1 reproduces the real product."""
    import edgpu
    from edgpu import model
    monkeypatch.setenv("EDGPU_KRON_SPLIT", "1"); monkeypatch.setenv("EDGPU_CSR_SPLIT", "1")
    m, nu, nd = model.half_filled_c(9)
    P = 4
    for r in (0, 2, 3):
        G = edgpu_mod.HvSector(m, nu, nd, storage=st, rank=r, size=P)
        c = edgpu.Communicator.emulated(r, P)
        G.attach_comm(c)
        q, d0 = model.dw_partition(G.DimDw, P, r)
        assert G.vecDim == G.DimUp * q
        mode, sb, rb = G.exchange_info()
        ld = (G.DimUp + 15) // 16 * 16      # rows of the explicit CSR's x-space travel with the line-aligned pitch
        assert mode == "needed-rows" and sb[r] == 0 and rb[r] == 0 and rb.sum() > 0 and (rb.sum() % (8 * G.DimUp) == 0 or rb.sum() % (8 * ld) == 0)
        v = np.random.RandomState(r).uniform(-0.5, 0.5, G.vecDim)
        hv = G.spHtimesV(v)
        assert np.all(np.isfinite(hv)) and np.abs(hv).max() > 0
        G.bench_begin(8); G.bench_steps(2); G.bench_steps(4, timed=True)
        ms, a, b = G.bench_result(4, ncoef=6)
        assert ms > 0 and np.all(np.isfinite(a)) and np.all(np.isfinite(b))
        G.close(); c.close()
    # one rank of one: nothing is emulated, the product is the real one
    S = oracle.Sector(m, nu, nd)
    G = edgpu_mod.HvSector(m, nu, nd, storage=st, rank=0, size=1, force_exchange=True)
    c = edgpu.Communicator.emulated(0, 1); G.attach_comm(c)
    v = np.random.RandomState(9).uniform(-0.5, 0.5, S.Dim)
    ref = S.spmatvec(v)
    assert np.abs(G.spHtimesV(v) - ref).max() <= 1e-12 * np.abs(ref).max()
    G.close(); c.close()


@pytest.mark.gpu
def test_rccl_code_path_single_rank(edgpu_mod):
    """edgpu_comm_unique_id/init_rccl + the split (local | exchange | dw) product and the device
    all-reduce, on a one-rank communicator: same numbers as the fused single-GPU path."""
    import cases
    from edgpu import model
    m, nu, nd = model.half_filled_c(7)
    v = cases.sinvec(4900)
    for st in ("kron", "direct", "csr"):
        A = edgpu_mod.HvSector(m, nu, nd, storage=st)
        B = edgpu_mod.HvSector(m, nu, nd, storage=st, force_exchange=True)
        B.comm_init_rccl(edgpu_mod.rccl_unique_id())
        ha, hb = A.spHtimesV(v), B.spHtimesV(v)
        assert np.abs(ha - hb).max() <= 1e-13 * np.abs(ha).max()
        ea, _, _ = A.sp_lanc_eigh(v, nitermax=200, threshold=1e-14)
        eb, _, _ = B.sp_lanc_eigh(v, nitermax=200, threshold=1e-14)
        assert abs(ea - eb) < 1e-11
        A.close(); B.close()


@pytest.mark.gpu
def test_torch_device_exchange_single_process():
    """TorchDeviceExchange views libedgpu's raw device pointers as torch tensors (world_size 1: the
    self-copy of allgatherv and the all-reduce run; peers are covered by the gloo variants).  Runs in
    a child process that imports torch before libedgpu.so, as bench.py does."""
    res = _launch("torchdev", "c_nb5_half", 1)
    assert res[0]["ok"]


@pytest.mark.gpu
def test_one_communicator_per_process_not_per_sector(edgpu_mod):
    """The reference sets MpiComm once (ED_VARS_GLOBAL.f90:332-347); a DMFT iteration then builds every (Nup,Ndw) sector.
    Here: ONE RCCL communicator, attached to all 49 sectors of the Ns=6 model run through the exchange code path
    (ground state of each against dense diagonalisation); the process has created exactly one communicator at the end."""
    from edgpu import model
    from oracle import oracle
    m, _, _ = model.half_filled_c(5)
    Ns = m["Ns"]
    n0 = edgpu_mod.comm_count()
    comm = edgpu_mod.Communicator.rccl(edgpu_mod.rccl_unique_id(), 0, 1)
    assert comm.allreduce_sum(np.ones(3)).tolist() == [1.0, 1.0, 1.0]
    for nup in range(Ns + 1):
        for ndw in range(Ns + 1):
            S = oracle.Sector(m, [nup], [ndw])
            w = np.linalg.eigvalsh(S.hmat())
            G = edgpu_mod.HvSector(m, [nup], [ndw], storage=("kron", "direct", "csr")[(nup + ndw) % 3], force_exchange=True)
            G.attach_comm(comm)
            assert G.exchange_info()[0] in ("needed-rows", "all-gather")
            e0, x, nl = G.sp_lanc_eigh(nitermax=min(300, S.Dim), threshold=1e-14)
            assert abs(e0 - w[0]) < 1e-9, (nup, ndw, e0, w[0])
            G.close()
    assert edgpu_mod.comm_count() == n0 + 1
    assert comm.sub(1) is comm
    comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [2, 3])
def test_multirank_sweep_with_one_communicator_and_shrunk_sectors(edgpu_mod, size):
    """world_size 2 and 3 on the one GPU: every rank creates ONE communicator object and sweeps over all sectors of a small
    model, including those with fewer dw columns than ranks, which run on the first DimDw ranks through a
    sub-communicator (ED_HAMILTONIAN.f90:65-93) while the other ranks sit the sector out."""
    res = _launch("sweep", "c_nb3_half", size, timeout=300)
    assert all(r["ok"] for r in res)
    assert res[0]["nsectors"] == 25 and res[0]["shrunk"] >= 2 and res[-1]["skipped"] >= 2
