"""world_size-2 gloo test of the N>1 path: band sharding + the single all-gather + resolve reproduce the
single-rank image bit for bit. The renderer behind the sharding logic is the CPU oracle here (no GPU in this
suite); on GPUs bench.py drives the same hobbyrenderer_amd.distributed functions with the HIP path and RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, spp, bounces, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hobbyrenderer_amd import native, scenes
    from hobbyrenderer_amd.distributed import render_sharded
    from oracle.binding import Oracle
    luts = native.precompute_atmosphere(2)
    sc, view, pos, _ = scenes.config_cornell(luts, w, h)
    o = Oracle(sc)
    acc = np.zeros((h, w, 4), np.float32)
    out = np.zeros((h, w, 4), np.float32)

    def render_band(y0, y1):
        for k in range(spp):
            o.render(scenes.fill_constants(view, pos, sc, k, bounces), acc, out, (0, y0, w, y1), nthreads=2)

    full = torch.from_numpy(acc)
    render_sharded(render_band, full, rank, world, lambda f, b: dist.all_gather_into_tensor(f, b))
    np.save(os.path.join(out_dir, f"acc_{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_band_sharding_matches_single_rank(tmp_path, luts):
    from hobbyrenderer_amd import scenes
    from oracle.binding import Oracle
    w, h, spp, bounces = 64, 36, 2, 4
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, w, h, spp, bounces, str(tmp_path)), nprocs=2, join=True)
    sc, view, pos, _ = scenes.config_cornell(luts, w, h)
    o = Oracle(sc)
    ref, _ = o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, bounces), w, h, spp)
    a0 = np.load(tmp_path / "acc_0.npy"); a1 = np.load(tmp_path / "acc_1.npy")
    assert np.array_equal(a0.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(a1.view(np.uint32), ref.view(np.uint32))


def _pipelined_worker(rank, world, port, w, h, bounces, out_dir, lanes=1, layout="rows"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hobbyrenderer_amd import native, scenes
    from hobbyrenderer_amd.distributed import PipelinedFrames, band_for_rank, column_view
    from oracle.binding import Oracle
    luts = native.precompute_atmosphere(2)
    sc, view, pos, _ = scenes.config_cornell(luts, w, h)
    o = Oracle(sc)
    accs = [np.zeros((h, w, 4), np.float32) for _ in range(lanes)]      # one accumulation image per lane (= per path-tracer context)
    out = np.zeros((h, w, 4), np.float32)
    state = {"first": 0}

    def band_renderer(acc):
        def render_band(y0, y1):   # frame f = accumulation index f alone, restarted from zero (three different frames)
            acc[y0:y1] = 0.0
            o.render(scenes.fill_constants(view, pos, sc, state["first"], bounces), acc, out, (0, y0, w, y1), nthreads=2)
            if layout == "columns":    # the oracle has no stripe mode: keep this rank's 8-pixel columns, clear the others
                cols = acc.reshape(h, w // 8, 8, 4)
                keep = np.zeros(w // 8, bool); keep[rank::world] = True
                cols[:, ~keep] = 0.0
        return render_band

    def resolve(a, b, stream):
        b.copy_(a / a[..., 3:4])

    y0, y1 = band_for_rank(h, world, rank)
    fulls = [torch.from_numpy(a) for a in accs]
    views = [column_view(f, world, rank) if layout == "columns" else f[y0:y1] for f in fulls]
    fused = {}
    if layout == "columns":     # the fused consumer (hrpt_resolve_columns_device on GPUs): shards -> output, plus the assembled image on request
        from hobbyrenderer_amd.distributed import columns_to_image

        def resolve_columns(shards, acc_out, out_img, stream):
            img = columns_to_image(shards, world, torch.empty((h, w, 4)))
            if acc_out is not None:
                acc_out.copy_(img)
            out_img.copy_(img / img[..., 3:4])
        fused = dict(resolve_columns=resolve_columns, keep_accumulation=True)
    frames = PipelinedFrames([band_renderer(a) for a in accs], views, h, w, rank, world,
                             lambda f, b: dist.all_gather_into_tensor(f, b), resolve, torch.device("cpu"), layout=layout, **fused)
    for f in range(3):
        state["first"] = f
        slot = frames.submit()
        assert slot == f % max(2, lanes)
        np.save(os.path.join(out_dir, f"frame{f}_{rank}.npy"), frames.gathered[slot].numpy().copy())
    frames.finish()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lanes,layout", [(1, "rows"), (2, "rows"), (2, "columns"), (4, "columns"), (3, "rows")])
def test_two_rank_pipelined_frames(tmp_path, luts, lanes, layout):
    """PipelinedFrames bookkeeping (double-buffered staging / gathered images; one or two lanes = contexts alternating frame by
    frame) under gloo, world 2: every frame of every rank equals the single-rank image of that accumulation index."""
    from hobbyrenderer_amd import scenes
    from oracle.binding import Oracle
    w, h, bounces = 48, 28, 3
    port = 30100 + (os.getpid() % 500)
    mp.spawn(_pipelined_worker, args=(2, port + lanes + (7 if layout == "columns" else 0), w, h, bounces, str(tmp_path), lanes, layout), nprocs=2, join=True)
    sc, view, pos, _ = scenes.config_cornell(luts, w, h)
    o = Oracle(sc)
    for f in range(3):
        acc = np.zeros((h, w, 4), np.float32); out = np.zeros((h, w, 4), np.float32)
        o.render(scenes.fill_constants(view, pos, sc, f, bounces), acc, out)
        for r in range(2):
            assert np.array_equal(np.load(tmp_path / f"frame{f}_{r}.npy").view(np.uint32), acc.view(np.uint32)), (f, r)


def test_band_for_rank():
    from hobbyrenderer_amd.distributed import band_for_rank
    assert [band_for_rank(1080, 8, r) for r in range(8)] == [(135 * r, 135 * (r + 1)) for r in range(8)]
    with pytest.raises(ValueError):
        band_for_rank(1080, 7, 0)
