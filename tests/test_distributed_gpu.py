"""GPU plumbing of the N>1 path that can be exercised on a one-GPU box: the zero-copy torch view of the
library-owned accumulation image and an RCCL (backend "nccl") all-gather through hobbyrenderer_amd.distributed
with world_size 1. The 2/4/8-GPU runs are the driver's (bench.py --gpus N)."""
import os

import numpy as np
import pytest

from hobbyrenderer_amd import scenes, structs as S

pytestmark = pytest.mark.gpu


def test_device_tensor_view_and_rccl_allgather(luts):
    import torch
    import torch.distributed as dist
    from hobbyrenderer_amd.distributed import device_tensor, render_sharded
    from hobbyrenderer_amd.native import PathTracerContext

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sc, view, pos, cfg = scenes.config_cornell(luts, 128, 72)
        ctx = PathTracerContext(0)
        ctx.upload_scene(sc); ctx.resize(128, 72)
        cb = scenes.fill_constants(view, pos, sc, 0, 4)
        accum_ptr, out_ptr = ctx.device_images()
        full = device_tensor(accum_ptr, (72, 128, 4), dev)

        # everything on torch's current stream (the default stream, handle 0): no host sync between render and collective
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

        def render_band(y0, y1):
            ctx.render(cb, accum_count=2, tile=(0, y0, 128, y1))

        # world 1 through the collective branch: band == whole image
        render_band(0, 72)
        band = full.clone()
        dist.all_gather_into_tensor(full, band)
        torch.cuda.synchronize(dev)
        ctx.resolve_output()
        host = ctx.read_accumulation()
        assert np.array_equal(full.cpu().numpy().view(np.uint32), host.view(np.uint32))
        assert (host[..., 3] == 2).all()
        y0, y1 = render_sharded(lambda a, b: None, full, 0, 1, dist.all_gather_into_tensor)
        assert (y0, y1) == (0, 72)

        # pipelined frames: gather + resolve of frame k on a second stream while frame k+1 renders (5 frames, 2 slots)
        from hobbyrenderer_amd.distributed import PipelinedFrames
        first = {"i": 0}

        def render_frame(a, b):
            ctx.render(scenes.fill_constants(view, pos, sc, first["i"], 4), accum_count=1, tile=(0, a, 128, b))

        frames = PipelinedFrames(render_frame, full[0:72], 72, 128, 0, 1, lambda f, b: dist.all_gather_into_tensor(f, b),
                                 lambda acc, out, stream: ctx.resolve_device(acc.data_ptr(), out.data_ptr(), 72 * 128, stream), dev)
        got = []
        for f in range(5):
            first["i"] = f
            slot = frames.submit()
            if f >= 3:
                got.append((f, slot))
        frames.finish()
        torch.cuda.synchronize(dev)
        for f, slot in got:          # the last two frames still sit in their slots
            ref_ctx = PathTracerContext(0)
            ref_ctx.upload_scene(sc); ref_ctx.resize(128, 72)
            ref_ctx.render(scenes.fill_constants(view, pos, sc, 0, 4), accum_count=f + 1)   # frames accumulate progressively: 0..f
            ra, ro = ref_ctx.read_accumulation(), ref_ctx.read_output()
            ref_ctx.close()
            assert np.array_equal(frames.gathered[slot].cpu().numpy().view(np.uint32), ra.view(np.uint32)), f
            assert np.array_equal(frames.output[slot].cpu().numpy().view(np.uint32), ro.view(np.uint32)), f
        # two lanes: frame k rendered by context k % 2 on its own stream (two frames in flight), gathers in frame order on the comm stream
        ctx2 = PathTracerContext(0)
        ctx2.upload_scene(sc); ctx2.resize(128, 72)
        lanes = [ctx, ctx2]
        streams = [torch.cuda.Stream(dev) for _ in lanes]
        for c, st in zip(lanes, streams):
            c.set_stream(st.cuda_stream)
        count = {"n": 1}

        def lane_render(c):
            return lambda a, b: c.render(scenes.fill_constants(view, pos, sc, 0, 4), accum_count=count["n"], tile=(0, a, 128, b))
        views = [device_tensor(c.device_images()[0], (72, 128, 4), dev)[0:72] for c in lanes]
        frames = PipelinedFrames([lane_render(c) for c in lanes], views, 72, 128, 0, 1, lambda f, b: dist.all_gather_into_tensor(f, b),
                                 lambda acc, out, stream: ctx.resolve_device(acc.data_ptr(), out.data_ptr(), 72 * 128, stream), dev, lane_streams=streams)
        got = []
        for f in range(6):
            count["n"] = f + 1                       # frame f = accumulation indices 0..f from scratch: every frame differs
            slot = frames.submit()
            if f >= 4:
                got.append((f, slot))
        frames.finish()
        torch.cuda.synchronize(dev)
        for f, slot in got:
            ref_ctx = PathTracerContext(0)
            ref_ctx.upload_scene(sc); ref_ctx.resize(128, 72)
            ref_ctx.render(scenes.fill_constants(view, pos, sc, 0, 4), accum_count=f + 1)
            ra, ro = ref_ctx.read_accumulation(), ref_ctx.read_output()
            ref_ctx.close()
            assert np.array_equal(frames.gathered[slot].cpu().numpy().view(np.uint32), ra.view(np.uint32)), f
            assert np.array_equal(frames.output[slot].cpu().numpy().view(np.uint32), ro.view(np.uint32)), f
        # the column layout through the same streams (world 1: every column is this rank's): stripes + strided staging copy + re-assembly
        from hobbyrenderer_amd.distributed import column_view

        def lane_render_cols(c):
            return lambda a, b: c.render(scenes.fill_constants(view, pos, sc, 0, 4), accum_count=count["n"], tile=(0, a, 128, b), stripes=(1, 0))
        cviews = [column_view(device_tensor(c.device_images()[0], (72, 128, 4), dev), 1, 0) for c in lanes]
        frames = PipelinedFrames([lane_render_cols(c) for c in lanes], cviews, 72, 128, 0, 1, lambda f, b: dist.all_gather_into_tensor(f, b),
                                 lambda acc, out, stream: ctx.resolve_device(acc.data_ptr(), out.data_ptr(), 72 * 128, stream), dev, lane_streams=streams,
                                 layout="columns",
                                 resolve_columns=lambda sh, acc, out, stream: ctx.resolve_columns_device(sh.data_ptr(), acc.data_ptr() if acc is not None else 0, out.data_ptr(), 128, 72, 1, stream))
        for f in range(3):
            count["n"] = f + 2
            slot = frames.submit()
        frames.finish()
        torch.cuda.synchronize(dev)
        ref_ctx = PathTracerContext(0)
        ref_ctx.upload_scene(sc); ref_ctx.resize(128, 72)
        ref_ctx.render(scenes.fill_constants(view, pos, sc, 0, 4), accum_count=4)
        assert np.array_equal(frames.gathered[slot].cpu().numpy().view(np.uint32), ref_ctx.read_accumulation().view(np.uint32))
        assert np.array_equal(frames.output[slot].cpu().numpy().view(np.uint32), ref_ctx.read_output().view(np.uint32))
        ref_ctx.close()
        ctx2.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("flags", [S.FRAME_DEFAULT, S.FRAME_MEGAKERNEL], ids=["wavefront", "megakernel"])
def test_column_interleaved_shards_assemble_to_the_full_image(luts, flags):
    """hrpt_render stripeCount / stripeIndex + distributed.column_view / columns_to_image: n ranks (here n contexts on one GPU) each
    render the 8-pixel columns k with k % n == rank; staging copies, a rank-major concatenation (what the all-gather delivers) and
    the re-assembly give the single-context image bit for bit. Widths that are not a multiple of 8 * n are refused."""
    import torch
    from hobbyrenderer_amd.distributed import column_view, columns_to_image, device_tensor
    from hobbyrenderer_amd.native import HrptError, PathTracerContext
    w, h, n, spp = 96, 40, 3, 2                                   # 12 columns of 8 pixels over 3 ranks
    sc, view, pos, cfg = scenes.config_cornell(luts, w, h)
    cb = scenes.fill_constants(view, pos, sc, 0, 4)
    dev = torch.device("cuda", 0)
    ref = PathTracerContext(0); ref.upload_scene(sc); ref.resize(w, h)
    ref.render(cb, accum_count=spp, flags=flags)
    want = ref.read_accumulation()
    shards = []
    for r in range(n):
        c = PathTracerContext(0); c.upload_scene(sc); c.resize(w, h)
        c.render(cb, accum_count=spp, flags=flags, stripes=(n, r))
        c.synchronize()
        img = device_tensor(c.device_images()[0], (h, w, 4), dev)
        mine = column_view(img, n, r)
        assert mine.shape == (h, w // 8 // n, 8, 4)
        host = c.read_accumulation().reshape(h, w // 8, 8, 4)
        others = np.delete(host, np.arange(r, w // 8, n), axis=1)
        assert (others == 0).all()                               # columns of other ranks are not touched
        shards.append(mine.contiguous())
        c.close()
    out = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
    columns_to_image(torch.stack(shards).reshape(-1), n, out)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # the fused consumer (hrpt_resolve_columns_device): shards -> assembled accumulation + resolved output in one pass, or output only
    stacked = torch.stack(shards).contiguous()
    acc2 = torch.zeros((h, w, 4), dtype=torch.float32, device=dev); res2 = torch.zeros_like(acc2); res3 = torch.zeros_like(acc2)
    ref.resolve_columns_device(stacked.data_ptr(), acc2.data_ptr(), res2.data_ptr(), w, h, n, torch.cuda.current_stream(dev).cuda_stream)
    ref.resolve_columns_device(stacked.data_ptr(), 0, res3.data_ptr(), w, h, n, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    assert np.array_equal(acc2.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(res2.cpu().numpy().view(np.uint32), ref.read_output().view(np.uint32)) and torch.equal(res2, res3)
    with pytest.raises(HrptError, match="multiple of 8"):
        ref.resolve_columns_device(stacked.data_ptr(), 0, res3.data_ptr(), w, h, 5, 0)
    with pytest.raises(HrptError, match="stripeIndex"):
        ref.render(cb, accum_count=1, stripes=(2, 2))
    with pytest.raises(ValueError):
        column_view(torch.empty((h, 100, 4)), n, 0)
    # a rectangle narrower than the stripe pattern: ranks beyond its columns render nothing and succeed
    ref.render(cb, accum_count=1, tile=(0, 0, 10, h), stripes=(4, 3))
    ref.close()


def test_in_process_allgather_of_bands(luts):
    """hrpt_allgather: n contexts inside one process (here all on GPU 0, so the peer copies are plain device copies), each renders
    its row band, the bands are exchanged context-to-context and every context ends with the full image == a single-context render."""
    from hobbyrenderer_amd import native
    from hobbyrenderer_amd.native import PathTracerContext
    w, h, n, spp = 96, 54, 3, 2
    sc, view, pos, cfg = scenes.config_cornell(luts, w, h)
    cb = scenes.fill_constants(view, pos, sc, 0, 4)
    ref = PathTracerContext(0); ref.upload_scene(sc); ref.resize(w, h)
    ref.render(cb, accum_count=spp)
    want_acc, want_out = ref.read_accumulation(), ref.read_output(); ref.close()
    ctxs = []
    try:
        for r in range(n):
            c = PathTracerContext(0); c.upload_scene(sc); c.resize(w, h); ctxs.append(c)
        rows = h // n
        for frame in range(2):                      # twice: the second exchange has to order itself behind the first resolve
            for r, c in enumerate(ctxs):
                c.render(cb, accum_count=spp, tile=(0, r * rows, w, (r + 1) * rows))
            native.allgather(ctxs)
            for c in ctxs:
                c.synchronize()
                assert np.array_equal(c.read_accumulation().view(np.uint32), want_acc.view(np.uint32))
                assert np.array_equal(c.read_output().view(np.uint32), want_out.view(np.uint32))
        with pytest.raises(native.HrptError):
            native.allgather([ctxs[0], ctxs[0]])
        ctxs[1].resize(w, h + 3)
        with pytest.raises(native.HrptError):
            native.allgather(ctxs)
    finally:
        for c in ctxs:
            c.close()
