"""Properties of the eddtra restatement (oracle/c/eddtra.c) that hold by construction of the
algorithm (phy/mod_eddtra.F90:228-1000); the restatement is otherwise unpinned (no reference build)."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.hostinit import step_indices


def _setup(cfg, **kw):
    from oracle.coracle import COracle
    case = make_case(cfg, **kw)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    orc = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(orc, case)
    return case, orc, dict(iu=iu, iv=iv)


@pytest.mark.parametrize("cfg", ["chan_s", "box_s"])
def test_zero_slopes_give_zero_transport(cfg):
    case, orc, _ = _setup(cfg, nslp0=0.0)
    six = step_indices(0, case.kdm)
    orc.stage("eddtra", *six)
    for nm in ("umfltd", "vmfltd", "utfltd", "usfltd"):
        a = orc.get(nm)[:, 4:-4, 4:-4]
        assert not np.any(a[np.isfinite(a)] != 0.0), nm


@pytest.mark.parametrize("cfg,nslp0", [("chan_s", 5e-3), ("chan_s", 0.5), ("box_s", 0.5)])
def test_gm_fluxes_bounded_and_column_neutral(cfg, nslp0):
    """No layer loses more than ffac = 1/16 of the mass available at the upstream point (:640-660),
    and the layer fluxes of a column are differences of interface fluxes that vanish at top and
    bottom, so they sum to ~0."""
    case, orc, m = _setup(cfg, nslp0=nslp0)
    kk = case.kdm
    m_, n_, mm, nn, k1m, k1n = step_indices(0, kk)
    orc.stage("eddtra", m_, n_, mm, nn, k1m, k1n)
    J, I = slice(4, 4 + case.jdm), slice(4, 4 + case.idm)
    scp2 = orc.get("scp2")[0]
    dp = orc.get("dp")[nn:nn + kk]
    for nm, msk, sh in (("umfltd", m["iu"], (0, -1)), ("vmfltd", m["iv"], (-1, 0))):
        f = orc.get(nm)[mm:mm + kk][:, J, I]
        wet = msk[J, I] > 0
        assert np.count_nonzero(f[:, wet]) > 0
        tot = f.sum(axis=0)[wet]
        scale = np.abs(f).sum(axis=0)[wet] + 1e-300
        assert np.all(np.abs(tot) <= 1e-9 * scale + 1e-3)
        # interior layers: |flux| <= ffac * max(epsilp, available thickness) * area on the side
        # that is being depleted; the larger of the two sides bounds both signs
        mass = dp * scp2
        ax = 2 if nm == "umfltd" else 1
        mass_a = np.roll(mass, shift=1, axis=ax)[:, J, I]          # (i-1,j) resp. (i,j-1)
        mass_b = mass[:, J, I]
        bound = 0.0625 * (np.maximum(mass_a, mass_b) + 1e-12 * scp2[J, I]) * (1 + 1e-12)
        assert np.all(np.abs(f[2:][:, wet]) <= bound[2:][:, wet])
