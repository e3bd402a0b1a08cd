"""hor3map on the GPU, through the C ABI (include/blomgpu_hor3map.h): bit-exact against the
committed golden vectors (the reference's own output), against the reference library where it
travelled with the snapshot (oracle/_ref/hor3map), and -- at the full channel slab size, 106 080
columns x 53 layers -- through size-independent properties (conservation, constants, agreement with
the small-slab results for identical columns)."""
import os

import numpy as np
import pytest

import h3m_cases as hc
from golden.make_hor3map_golden import NCOL, N_SRC, N_DST, N_GRD, golden_cases

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_gpu_matches_golden():
    golden = np.load(os.path.join(HERE, "golden", "hor3map.npz"))
    bad = []
    for name, cfg, kind, rm, seed, dec in golden_cases():
        x, u, xd, ug = hc.make_columns(seed, NCOL, N_SRC, N_DST, N_GRD, kind, dec)
        g = {k: golden[f"{name}/{k}"] for k in ("polycoeff", "u_dst", "x_grd", "errs", "n_act", "m_act")}
        bad += hc.compare(g, hc.run_gpu(*cfg, x, u, xd, ug, rm), name)
    assert not bad, "\n".join(bad[:10])


@pytest.mark.parametrize("cfg", hc.CONFIGS, ids=lambda c: "-".join(map(str, map(int, c))))
def test_gpu_matches_reference(cfg):
    if not hc.have_ref():
        pytest.skip("oracle/_ref/hor3map/libhor3mapref.so did not travel")
    bad = []
    for ik, kind in enumerate(hc.KINDS):
        for rm in (hc.METHOD_1, hc.METHOD_2):
            dec = (ik + rm) % 2 == 0
            x, u, xd, ug = hc.make_columns(177 + ik, 1500, 53, 53, 54, kind, dec)
            bad += hc.compare(hc.run_ref(*cfg, x, u, xd, ug, rm), hc.run_gpu(*cfg, x, u, xd, ug, rm),
                              f"{kind} rm{rm} dec{dec}")
    assert not bad, "\n".join(bad[:10])


def test_errors_are_reported_like_the_reference():
    from blom_amd import hor3map as h3
    x, u, xd, ug = hc.make_columns(3, 64, 12, 9, 5, "bad")
    g = h3.ReconGrid(64, 12, h3.PPM, 6, 4)
    with pytest.raises(h3.Hor3mapError, match="Source grid edges do not monotonically"):
        g.prepare_reconstruction(x)
    s = h3.ReconSrc(g, h3.MONOTONIC, False, False)
    with pytest.raises(h3.Hor3mapError, match="Call 'prepare_reconstruction' first!"):
        s.reconstruct(u)
    g.free()
    with pytest.raises(h3.Hor3mapError, match="Invalid reconstruction method!"):
        h3.ReconGrid(8, 5, 99)
    g = h3.ReconGrid(64, 12, h3.PLM)
    x2, u2, _, _ = hc.make_columns(4, 64, 12, 9, 5, "ocean")
    g.prepare_reconstruction(x2)
    s = h3.ReconSrc(g, 777, False, False)
    with pytest.raises(h3.Hor3mapError, match="Invalid limiting method for PLM!"):
        s.reconstruct(u2)
    s = h3.ReconSrc(g, h3.MONOTONIC, False, False)
    s.reconstruct(u2)
    with pytest.raises(h3.Hor3mapError, match="Invalid regrid method!"):
        s.regrid(ug, hc.MISSING, 5)
    g.free()


def test_full_channel_slab_properties():
    """106 080 columns (the channel's wet points) x 53 layers, BLOM's tracer configuration"""
    from blom_amd import hor3map as h3
    ncol, n = 106080, 53
    x, u, xd, ug = hc.make_columns(11, ncol, n, n, 8, "tracer")
    cfg = (hc.PPM, 6, 4, hc.NON_OSCILLATORY_POSDEF, True, False)
    g = h3.ReconGrid(ncol, n, cfg[0], cfg[1], cfg[2])
    s = h3.ReconSrc(g, cfg[3], cfg[4], cfg[5])
    r = h3.Remap(g, n)
    g.prepare_reconstruction(x)
    s.reconstruct(u)
    r.prepare_remapping(xd)
    ud = r.remap(s)
    src = (u * np.diff(x, axis=1)).sum(1)
    dst = (ud * np.diff(xd, axis=1)).sum(1)
    assert np.all(np.abs(src - dst) <= 1e-12 * np.abs(u).max() * x[:, -1])
    # positive definite limiting: non-negative sources stay non-negative
    sp = h3.ReconSrc(g, cfg[3], cfg[4], cfg[5])
    sp.reconstruct(np.abs(u))
    up = r.remap(sp)
    assert up.min() >= -1e-13 * np.abs(u).max(), up.min()
    # remapping onto the source grid itself returns the source means (columns without thin layers,
    # where no cells are merged)
    rng = np.random.default_rng(1)
    xs = np.concatenate([np.zeros((ncol, 1)), np.cumsum(rng.uniform(0.5, 2.0, (ncol, n)), 1)], 1)
    g.prepare_reconstruction(xs)
    s.reconstruct(u)
    r.prepare_remapping(xs)
    assert np.all(np.abs(r.remap(s) - u) <= 1e-12 * np.abs(u).max())
    g.prepare_reconstruction(x)
    s.reconstruct(u)
    # the first 1500 columns give the same bits as a 1500-column slab
    small = hc.run_gpu(*cfg, x[:1500].copy(), u[:1500].copy(), xd[:1500].copy(), ug[:1500].copy(), hc.METHOD_1)
    r.prepare_remapping(xd)
    assert np.array_equal(r.remap(s)[:1500], small["u_dst"])
    g.free()


def test_many_fields_in_one_launch_equal_single_calls():
    """reconstruct_many / remap_many (the tracer loop in one launch) against one call per field"""
    from blom_amd import hor3map as h3
    ncol, n, nd = 3000, 53, 40
    x, u, xd, ug = hc.make_columns(21, ncol, n, nd, 8, "tracer")
    rng = np.random.default_rng(2)
    fields = [np.ascontiguousarray(u * s + o) for s, o in ((1.0, 0.0), (-0.5, 1.0), (2.0, 3.0), (0.1, -0.2), (1.5, 0.5))]
    fields.append(np.ascontiguousarray(rng.standard_normal((ncol, n))))
    lims = [h3.NON_OSCILLATORY_POSDEF, h3.MONOTONIC, h3.NON_OSCILLATORY, h3.NO_LIMITING, h3.MONOTONIC,
            h3.NON_OSCILLATORY_POSDEF]
    for method in (h3.PPM, h3.PQM, h3.PLM):
        g = h3.ReconGrid(ncol, n, method, 6, 4)
        g.prepare_reconstruction(x)
        r = h3.Remap(g, nd)
        r.prepare_remapping(xd)
        singles = []
        for f, lim in zip(fields, lims):
            s = h3.ReconSrc(g, lim, True, False)
            s.reconstruct(f)
            singles.append((s.extract_polycoeff(), r.remap(s)))
        srcs = [h3.ReconSrc(g, lim, True, False) for lim in lims]
        h3.reconstruct_many(g, srcs, fields)
        outs = h3.remap_many(srcs, r)
        for k, s in enumerate(srcs):
            assert np.array_equal(s.extract_polycoeff(), singles[k][0]), (method, k)
            assert np.array_equal(outs[k], singles[k][1]), (method, k)
        g.free()
