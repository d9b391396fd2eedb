"""Shared model zoo of the parity tests (inputs only)."""
import json
import os

import numpy as np

from edgpu import model

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "survey_known_answers.json")


def sinvec(n):
    """v(i)=sin(i), i=1..N -- the deterministic vector of SURVEY 8c."""
    return np.sin(np.arange(1, n + 1, dtype=np.float64))


def sigma_z(Nspin=1):
    O = np.zeros((Nspin, Nspin, 2, 2))
    for s in range(Nspin):
        O[s, s, 0, 0] = 1.0
        O[s, s, 1, 1] = -1.0
    return O


def model_from_golden(g):
    gm = dict(g["model"])
    kw = dict(Norb=gm["Norb"], Nbath=gm["Nbath"], bath_type=gm["bath_type"], Uloc=tuple(gm["Uloc"]),
              xmu=gm.get("xmu", 0.0), hfmode=gm.get("hfmode", True), Jh=gm.get("Jh", 0.0), Ust=gm.get("Ust", 0.0),
              Jx=gm.get("Jx", 0.0), Jp=gm.get("Jp", 0.0), ed_total_ud=gm.get("ed_total_ud", True))
    if gm.get("sigma_z_basis"):
        kw.update(H_basis=[sigma_z()], lambda_imp=gm["lambda_imp"])
    if "impHloc_diag" in gm:
        imp = np.zeros((1, 1, gm["Norb"], gm["Norb"]))
        for a, x in enumerate(gm["impHloc_diag"]):
            imp[0, 0, a, a] = x
        kw["impHloc"] = imp
    return model.make_model(**kw)


def golden_cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


def zoo():
    """(name, model, nups, ndws) covering the operator variants of SURVEY 8a."""
    Z = []
    for nb in (1, 2, 3, 5, 7):
        m, nu, nd = model.half_filled_c(nb)
        Z.append(("c_nb%d_half" % nb, m, nu, nd))
    m = model.make_model(Norb=1, Nbath=5, Uloc=(3.0,), xmu=0.3, hfmode=False)
    Z.append(("nb5_offhalf_2_4", m, [2], [4]))
    Z.append(("nb5_empty_up", m, [0], [3]))
    Z.append(("nb5_full", m, [6], [6]))
    Z.append(("nb5_one_each", m, [1], [1]))
    m = model.make_model(Norb=1, Nbath=6, Uloc=(2.0,))   # even Nbath branch of the default bath
    Z.append(("nb6_even", m, [3], [4]))
    m = model.make_model(Norb=1, Nspin=2, Nbath=4, Uloc=(2.0,), bath_e=np.random.RandomState(1).uniform(-1, 1, (2, 1, 4)),
                         bath_v=np.random.RandomState(2).uniform(0.2, 0.8, (2, 1, 4)))
    Z.append(("nspin2_random_bath", m, [2], [3]))
    imp = np.zeros((1, 1, 2, 2)); imp[0, 0, 0, 0] = 0.2; imp[0, 0, 1, 1] = -0.1; imp[0, 0, 0, 1] = imp[0, 0, 1, 0] = 0.35
    m = model.make_model(Norb=2, Nbath=2, Uloc=(2.0, 1.5), Ust=1.2, Jh=0.2, impHloc=imp)
    Z.append(("norb2_normal_offdiag_hloc", m, [3], [3]))
    m = model.make_model(Norb=2, Nbath=2, Uloc=(2.0, 1.5), Ust=1.2, Jh=0.2, Jx=0.2, Jp=0.15, impHloc=imp)
    Z.append(("norb2_normal_JxJp", m, [3], [2]))
    m = model.make_model(Norb=2, Nbath=3, bath_type="hybrid", Uloc=(2.0, 2.0), Ust=1.0, Jh=0.3, Jx=0.3, Jp=0.3,
                         bath_v=np.random.RandomState(3).uniform(0.2, 0.6, (1, 2, 3)))
    Z.append(("norb2_hybrid_JxJp", m, [2], [3]))
    sx = np.zeros((1, 1, 2, 2)); sx[0, 0, 0, 1] = sx[0, 0, 1, 0] = 1.0
    m = model.make_model(Norb=2, Nbath=2, bath_type="replica", H_basis=[sigma_z(), sx], lambda_imp=[1.0, 0.3],
                         Uloc=(2.0, 2.0), Jh=0.25, Ust=1.5, Jx=0.25, Jp=0.25)
    Z.append(("replica_sz_sx_JxJp", m, [3], [3]))
    m = model.make_model(Norb=3, Nbath=1, Uloc=(2.0, 1.8, 1.6), Ust=1.1, Jh=0.15, Jx=0.15, Jp=0.15)
    Z.append(("norb3_nb1_JxJp", m, [3], [3]))
    m = model.make_model(Norb=2, Nbath=3, ed_total_ud=False, Uloc=(2.0, 2.0), Ust=1.5, Jh=0.25, impHloc=np.diag([0.3, -0.3]).reshape(1, 1, 2, 2))
    Z.append(("orbs_norb2_nb3", m, [2, 2], [2, 2]))
    Z.append(("orbs_norb2_nb3_uneven", m, [1, 3], [2, 0]))
    m = model.make_model(Norb=2, Nbath=4, ed_total_ud=False, Uloc=(2.0, 1.0), Ust=0.8, Jh=0.1, impHloc=np.diag([0.1, -0.2]).reshape(1, 1, 2, 2),
                         bath_v=np.random.RandomState(8).uniform(0.2, 0.6, (1, 2, 4)), bath_e=np.random.RandomState(9).uniform(-1, 1, (1, 2, 4)))
    Z.append(("orbs_norb2_nb4", m, [2, 3], [3, 2]))
    m = model.make_model(Norb=1, Nbath=3, Nph=3, w0_ph=0.7, g_ph=(0.4,), Uloc=(2.0,))
    Z.append(("phonons_nb3_nph3", m, [2], [2]))
    m = model.make_model(Norb=2, Nbath=2, Nph=2, w0_ph=0.5, g_ph=(0.3, 0.2), Uloc=(2.0, 2.0), Ust=1.0, Jh=0.2, Jx=0.2, Jp=0.2)
    Z.append(("phonons_norb2_JxJp", m, [3], [3]))
    m = model.make_model(Norb=2, Nbath=2, ed_total_ud=False, Nph=2, w0_ph=0.5, g_ph=(0.3, 0.2), Uloc=(2.0, 2.0), Ust=1.0, Jh=0.2)
    Z.append(("phonons_orbs", m, [1, 2], [2, 1]))
    return Z
