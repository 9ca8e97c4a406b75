"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical Scene inputs.

Bar (BASELINE.json north_star): accumulated radiance within 1e-4 relative L2 per pixel. Because both
sides implement the numeric contract of include/hobbyrt/detmath.h with a fixed expression order, the
tests first assert BIT equality and report the 1e-4 figure as the fallback bar.
"""
import numpy as np
import pytest

from conftest import rel_l2_per_pixel
from hobbyrenderer_amd import scenes, structs as S

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: relative L2 per pixel


@pytest.fixture(scope="module")
def ctx():
    from hobbyrenderer_amd.native import PathTracerContext
    c = PathTracerContext(0)
    yield c
    c.close()


def _run_both(ctx, sc, view, pos, w, h, spp, bounces, flags, first=0):
    from oracle.binding import Oracle, OrStats
    ctx.upload_scene(sc)
    ctx.resize(w, h)
    ctx.reset_stats()
    ctx.render(scenes.fill_constants(view, pos, sc, first, bounces), accum_count=spp, flags=flags)
    acc = ctx.read_accumulation()
    out = ctx.read_output()
    st = ctx.stats()
    o = Oracle(sc)
    ost = OrStats()
    oacc, oout = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, bounces), w, h, spp, first_index=first, stats=ost)
    o.close()
    return acc, out, st, oacc, oout, ost


def _assert_parity(acc, out, st, oacc, oout, ost):
    assert st.closestRays == ost.closestRays and st.shadowRays == ost.shadowRays and st.paths == ost.paths
    err = rel_l2_per_pixel(acc, oacc)
    assert err.max() <= TOL, f"max per-pixel rel L2 {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    assert np.array_equal(acc.view(np.uint32), oacc.view(np.uint32)), f"not bit-exact: {np.count_nonzero(acc != oacc)} floats differ (max rel-L2 {err.max():.3e})"
    assert np.array_equal(out.view(np.uint32), oout.view(np.uint32))


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_config1_default_cube(ctx, luts, flags):
    """BASELINE config 1: ProceduralDefaultCube 256x256, 1 spp, 1 bounce."""
    sc, view, pos, cfg = scenes.config_cube(luts, 256)
    _assert_parity(*_run_both(ctx, sc, view, pos, 256, 256, cfg["spp"], cfg["max_bounces"], flags))


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_cube_multibounce_accumulated(ctx, luts, flags):
    sc, view, pos, _ = scenes.config_cube(luts, 96)
    _assert_parity(*_run_both(ctx, sc, view, pos, 96, 96, 3, 5, flags))


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_config2_cornell_reduced(ctx, luts, flags):
    """BASELINE config 2 scene at 320x180 (the oracle finishes in seconds), 8 spp, 4 bounces."""
    sc, view, pos, cfg = scenes.config_cornell(luts, 320, 180)
    _assert_parity(*_run_both(ctx, sc, view, pos, 320, 180, cfg["spp"], cfg["max_bounces"], flags))


@pytest.mark.parametrize("flags", [S.FRAME_MEGAKERNEL, S.FRAME_DEFAULT], ids=["megakernel", "default"])
def test_cornell_point_and_spot_lights(ctx, luts, flags):
    sc, view, pos, cfg = scenes.config_cornell(luts, 160, 90, extra_lights=True)
    _assert_parity(*_run_both(ctx, sc, view, pos, 160, 90, 4, 6, flags))
