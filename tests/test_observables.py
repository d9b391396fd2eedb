"""local observables from the device-side impurity-configuration weights against the
oracle's restatement of lanc_observables (ED_OBSERVABLES.f90:150-215)."""
import numpy as np
import pytest

import cases
from oracle import oracle

pytestmark = pytest.mark.gpu
SEL = [z for z in cases.zoo() if z[0] in ("c_nb5_half", "nb5_offhalf_2_4", "norb2_normal_JxJp", "norb3_nb1_JxJp", "orbs_norb2_nb3", "phonons_norb2_JxJp", "phonons_orbs")]


@pytest.mark.parametrize("name,m,nu,nd", SEL, ids=[z[0] for z in SEL])
def test_observables_match_oracle(edgpu_mod, name, m, nu, nd):
    S = oracle.Sector(m, nu, nd)
    w, U = np.linalg.eigh(S.hmat())
    gs = U[:, 0]
    ref = oracle.lanc_observables(S, gs)
    G = edgpu_mod.HvSector(m, nu, nd, storage="direct")
    got = G.observables(gs)
    for k, v in ref.items():
        np.testing.assert_allclose(got[k], v, rtol=0, atol=1e-13, err_msg=k)
    assert abs(got["Prob"].sum() - 1.0) < 1e-13
    # same from a device-resident vector (no PCIe copy of the state)
    dv = edgpu_mod.DeviceVector(S.Dim).upload(gs)
    np.testing.assert_allclose(G.observables(dv)["docc"], ref["docc"], atol=1e-13)
    G.close()


@pytest.mark.parametrize("name,m,nu,nd", [z for z in cases.zoo() if z[0] in ("nb5_offhalf_2_4", "norb2_normal_JxJp", "orbs_norb2_nb3_uneven", "phonons_orbs", "nb6_even")],
                         ids=["nb5_offhalf_2_4", "nb6_even", "norb2_normal_JxJp", "orbs_norb2_nb3_uneven", "phonons_orbs"])
def test_twin_sector_vector(edgpu_mod, name, m, nu, nd):
    """vector_twin(i) = cvec(Order(i)) with Order from the oracle's restatement of twin_sector_order; and the
    reordered vector is an eigenvector of the twin sector when the model is spin symmetric."""
    S = oracle.Sector(m, nu, nd)
    v = np.random.RandomState(9).uniform(-1, 1, S.Dim)
    order = oracle.twin_sector_order(S)
    G = edgpu_mod.HvSector(m, nu, nd, storage="direct")
    np.testing.assert_array_equal(G.twin_reorder(v), v[order - 1])
    # the same with the state resident in HBM (device pointers in and out, nothing crosses PCIe inside the call)
    dv = edgpu_mod.DeviceVector(S.Dim).upload(v)
    dt = G.twin_reorder(dv)
    np.testing.assert_array_equal(dt.download(), v[order - 1])
    dt.free(); dv.free()
    G.close()


@pytest.mark.parametrize("case", ["norb2_normal_offdiag_hloc", "norb2_normal_JxJp", "norb3_nb1_JxJp", "phonons_norb2_JxJp"])
def test_offdiagonal_impurity_density_matrix(edgpu_mod, case):
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    w, U = np.linalg.eigh(S.hmat()); gs = U[:, 0]
    ref = oracle.imp_density_offdiag(S, gs)
    G = edgpu_mod.HvSector(m, nu, nd, storage="direct")
    np.testing.assert_allclose(G.imp_density_offdiag(gs), ref, rtol=0, atol=1e-13)
    G.close()


LE = ["c_nb5_half", "nb5_offhalf_2_4", "norb2_normal_offdiag_hloc", "norb2_normal_JxJp", "norb2_hybrid_JxJp", "replica_sz_sx_JxJp", "norb3_nb1_JxJp",
      "orbs_norb2_nb3", "phonons_norb2_JxJp", "phonons_orbs"]


@pytest.mark.parametrize("case", LE)
def test_local_energy_terms(edgpu_mod, case):
    """edgpu_local_energy against the oracle's restatement of lanc_local_energy (ED_OBSERVABLES.f90:372-616): Epot (with the
    Hartree shift), Ehartree, Eknot (diagonal + off-diagonal impHloc), Dust, Dund and -- the off-diagonal pass -- the
    spin-exchange and pair-hopping expectation values Dse, Dph; ground state and a random state, host and device input."""
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    w, U = np.linalg.eigh(S.hmat())
    rnd = np.random.RandomState(12).uniform(-1, 1, S.Dim); rnd /= np.linalg.norm(rnd)
    for st in ("kron", "direct"):
        G = edgpu_mod.HvSector(m, nu, nd, storage=st)
        for vec in (U[:, 0], rnd):
            ref = oracle.local_energy(S, vec, peso=0.7)
            got = G.local_energy(vec, peso=0.7)
            for k in ref:
                assert abs(got[k] - ref[k]) <= 1e-12 * max(1.0, abs(ref[k])), (st, k, got[k], ref[k])
        dv = edgpu_mod.DeviceVector(S.Dim).upload(rnd)
        got = G.local_energy(dv)
        ref = oracle.local_energy(S, rnd)
        assert all(abs(got[k] - ref[k]) <= 1e-12 * max(1.0, abs(ref[k])) for k in ref)
        if m["Norb"] > 1 and (m.get("Jx", 0) or m.get("Jp", 0)) and m.get("ed_total_ud", True):
            assert abs(ref["Dse"]) + abs(ref["Dph"]) > 1e-6       # the off-diagonal pass is really exercised
        G.close()
    # <H> of an eigenstate: Eknot + (Epot) + bath/hybridisation parts is not available here, but the interaction part must be
    # consistent with the density terms: Epot - Ehartree - Jx Dse - Jp Dph = U/Ust/(Ust-Jh) weighted densities
    ref = oracle.local_energy(S, U[:, 0])
    dens = ref["Epot"] - ref["Ehartree"] - m.get("Jx", 0.0) * ref["Dse"] - m.get("Jp", 0.0) * ref["Dph"]
    assert dens >= -1e-12
