"""The Bruneton LUT producer (hobbyrenderer_amd/csrc/atmosphere_precompute.cpp; stand-in for the reference's bin/bruneton/*.dat,
src/CommonResources.cpp:519-569, constants src/shaders/Atmosphere.hlsli:41-75). The LUT CONTENTS are unpinned by the reference (its files
and their producer are not in the tree; the number of scattering orders they hold is unknown): these tests check the physics of the
tables (energy per order, ranges) and -- the property the parity tests rely on -- that host threads and the GPU produce the same bits."""
import ctypes as C

import numpy as np
import pytest

from hobbyrenderer_amd import native, structs as S

NS, NI = 32 * 128 * 256, 16 * 64


def _tables(orders, device=-1):
    return native.precompute_atmosphere(orders=orders, device=device, cache=False)


def test_single_scattering_only_is_the_round_2_table():
    """orders = 1: transmittance + single scattering, zero irradiance -- the tables of rounds 1 and 2, kept as an option."""
    t, s, i = _tables(1)
    assert not i[..., :3].any() and (i[..., 3] == 1).all()
    assert (t[..., :3] > 0).all() and (t[..., :3] <= 1).all() and (s >= 0).all() and np.isfinite(s).all()
    # sun at the zenith seen from the ground looking up: more blue than red (Rayleigh), Mie term in alpha
    assert s[0, 127, 31, 2] > s[0, 127, 31, 0] > 0 and s[0, 127, 31, 3] > 0


def test_energy_grows_with_every_order_and_converges(luts):
    """Every scattering order adds light everywhere (never removes any), less than the one before: sum(order 2) / sum(single) ~ 1.27, the
    default four orders ~ 1.36. The single-Mie alpha channel and the transmittance do not depend on the order count."""
    t1, s1, _ = _tables(1)
    t2, s2, i2 = _tables(2)
    t4, s4, i4 = luts                                  # the default tables (4 orders, cached)
    assert np.array_equal(t1, t2) and np.array_equal(t1, t4)
    assert np.array_equal(s1[..., 3], s2[..., 3]) and np.array_equal(s1[..., 3], s4[..., 3])
    assert (s2[..., :3] >= s1[..., :3]).all() and (s4[..., :3] >= s2[..., :3]).all()
    e1, e2, e4 = (float(x[..., :3].astype(np.float64).sum()) for x in (s1, s2, s4))
    assert 1.15 < e2 / e1 < 1.40 and e2 < e4 < e2 * 1.15
    assert (e4 - e2) < (e2 - e1)                       # orders 3 + 4 together add less than order 2 alone
    # ground irradiance from the sky (orders >= 2 of the Bruneton scheme): positive under a high sun, ~0.2 W/m2/nm-scale units, growing with the orders
    assert (i2[..., :3] >= 0).all() and (i4[..., :3] >= i2[..., :3]).all()
    assert 0.05 < float(i4[0, 63, 2]) < 1.0 and float(i4[0, 63, 2]) > float(i4[0, 63, 0])       # ground level, sun at the zenith: bluish sky light
    assert float(i4[0, 0, :3].max()) < 0.02            # sun far below the horizon


def test_zenith_sky_radiance_is_in_the_expected_range(luts):
    """Zenith radiance at ground level, sun 45 degrees up (the default directional light, src/Scene.cpp:643-665), unit intensity, in the
    table's radiometric units (solar irradiance 1.47 / 1.85 / 1.91 W/m2/nm): blue 0.03-0.12 (measured 0.048), red 0.1-0.3 of blue (0.15) -- and
    10-60 % above what single scattering alone gives (red + 18 %, green + 27 %, blue + 46 %: the higher orders matter most where the optical depth is largest)."""
    from hobbyrenderer_amd import scenes
    from oracle.binding import Oracle
    sun = np.array([0.0, 0.70710677, -0.70710677], np.float32)
    out = {}
    for name, tables in (("four", luts), ("single", _tables(1))):
        sc = scenes.config_cube(tables, 16)[0]
        o = Oracle(sc)
        out[name] = o.sky_radiance((0, 0, 0), (0, 1, 0), sun, 1.0, False)
        o.close()
    z = out["four"]
    assert 0.03 < z[2] < 0.12 and 0.1 < z[0] / z[2] < 0.3
    gain = out["four"] / out["single"]
    assert (gain > 1.10).all() and (gain < 1.60).all() and gain[2] > gain[1] > gain[0]


def _synthetic_inputs(seed):
    """Smooth positive tables of the right shapes: inputs for one pass on both executors (the bits must agree whatever the inputs are)."""
    rng = np.random.default_rng(seed)
    t = native.precompute_atmosphere(orders=1, device=-1, cache=False)[0]
    def tab(n, scale):
        return (rng.random((n, 3)).astype(np.float32) * np.float32(scale) + np.float32(scale * 0.1))
    return {"T": t, "dIrr": tab(NI, 0.5), "dR": tab(NS, 0.02), "dM": tab(NS, 0.01), "dDens": tab(NS, 1e-3), "dMulti": tab(NS, 5e-3), "scat": rng.random((NS, 4)).astype(np.float32)}


def _run_pass(inp, pass_id, order, first, count, device):
    out = np.zeros((NI if pass_id == 4 else NS, 3), np.float32)
    scat = inp["scat"].copy()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = native.lib.hrpt_atmosphere_pass(pass_id, order, first, count, p(inp["T"]), p(inp["dIrr"]), p(inp["dR"]), p(inp["dM"]), p(inp["dDens"]), p(inp["dMulti"]), p(scat), p(out), 0, device)
    assert rc == 0
    return out, scat


def test_pass_hook_argument_checks():
    inp = _synthetic_inputs(1)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    out = np.zeros((NS, 3), np.float32)
    args = (p(inp["T"]), p(inp["dIrr"]), p(inp["dR"]), p(inp["dM"]), p(inp["dDens"]), p(inp["dMulti"]), p(inp["scat"]), p(out), 0, -1)
    assert native.lib.hrpt_atmosphere_pass(2, 2, 0, 1, *args) == -1            # not a hooked pass
    assert native.lib.hrpt_atmosphere_pass(3, 2, NS - 1, 2, *args) == -1       # range past the table
    assert native.lib.hrpt_atmosphere_pass(4, 1, 0, NI, *args) == 0


@pytest.mark.gpu
def test_gpu_and_host_threads_compute_the_same_bits_pass_by_pass():
    """Scattering density (orders 2 and 3: with and without the single-scattering phase functions), indirect irradiance and multiple scattering
    on a few thousand texels spread over the tables, from identical inputs: every float identical."""
    inp = _synthetic_inputs(7)
    for pass_id, order, first, count in ((3, 2, 0, 1500), (3, 3, 517_000, 1500), (3, 2, NS - 1200, 1200), (4, 1, 0, NI), (4, 2, 0, NI), (5, 2, 0, 20000), (5, 3, 777_777, 20000)):
        host, hscat = _run_pass(inp, pass_id, order, first, count, -1)
        dev, dscat = _run_pass(inp, pass_id, order, first, count, 0)
        assert np.array_equal(host.view(np.uint32), dev.view(np.uint32)), (pass_id, order, first)
        assert np.array_equal(hscat.view(np.uint32), dscat.view(np.uint32)), (pass_id, order, first)
        assert host[first:first + count].any()


@pytest.mark.gpu
def test_gpu_tables_equal_the_host_tables(luts):
    """The whole precomputation on the GPU (what a box with a GPU and no cache file runs) against the session's tables: identical arrays. The
    session tables come from the cache when this checkout has one (computed by host threads here: the cache file travels with the repository
    snapshot), else they were just computed on this GPU -- then two orders by host threads are compared instead."""
    import os
    t, s, i = _tables(native.ATMOSPHERE_ORDERS, device=0)
    assert np.array_equal(t, luts[0]) and np.array_equal(s.view(np.uint32), luts[1].view(np.uint32)) and np.array_equal(i, luts[2])
    if not os.path.exists(native._atmosphere_cache_path(native.ATMOSPHERE_ORDERS)):
        th, sh, ih = _tables(2, device=-1)
        tg, sg, ig = _tables(2, device=0)
        assert np.array_equal(th, tg) and np.array_equal(sh.view(np.uint32), sg.view(np.uint32)) and np.array_equal(ih, ig)
