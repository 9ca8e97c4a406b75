#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: Mrays/s at 1920x1080, 8 spp, 4 bounces (config 2: Cornell-box-class
procedural scene) on N MI355X of one node.

A "step" is one full render of that frame through the C ABI (hrpt_render with accumCount = spp) with the
Scene, BVH and LUTs already resident in HBM. N > 1: one process per GPU (torch.distributed over RCCL), the
image is sharded by row bands (pixels are independent: RNG.hlsli:21-27), scene replicated, one all-gather of
the RGBA32F accumulation bands per step, then the resolve (Output = accum.rgb / accum.a). Total work is fixed
as N grows ("strong" scaling); value is the whole-job aggregate.

Rays = closest-hit queries + shadow queries actually launched (device counters; identical to the oracle's).
roofline: algorithmic bytes per ray from SURVEY.md 8(d) -- B_closest = 768 + 32 n + 48 t, B_shadow = 36 + 32 n
+ 48 t, + 48 B per pixel per spp -- with n (AABB tests) and t (triangle tests) counted by the CPU oracle on a
bounded sample of the same workload, divided by the dominant kernel's device time (HIP events recorded by the
library on its own stream). cpu_baseline: the oracle (a port; the reference has no CPU path) timed on the host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def _pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed under
    profiles/ (FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def _pmc_valu(kernel):
    """VALU-issue counters of `kernel` on config 2 from the committed rocprofv3 --pmc passes (profiles/r*_pmc_valu_config2.json):
    the kernels of this path are VALU-issue-bound, so these say more about them than the HBM fraction does."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_valu_config2.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return json.load(f).get(kernel)
    except (OSError, ValueError):
        return None


def _baseline_metric():
    """The metric string exactly as BASELINE.json spells it (the file travels with the repository)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "Mrays/s at 1920\u00d71080, 8 spp, 4 bounces; 1/2/4/8 GPU scaling"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--mode", choices=["default", "megakernel", "wavefront"], default="default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal only: all ranks share GPU 0 and the all-gather runs over gloo through host memory; the printed value is NOT a result")
    ap.add_argument("--serial-gather", action="store_true", help="N>1: all-gather in stream order after each render instead of overlapping it with the next frame")
    ap.add_argument("--frames-in-flight", type=int, default=2, choices=(1, 2),
                    help="independent steps (frames) in flight on each GPU: 2 = consecutive steps alternate between two path-tracer contexts "
                         "(own streams, queues and images) so that one frame's kernel tails overlap the other's; 1 = one frame at a time")
    ap.add_argument("--sharding", choices=("columns", "rows"), default="columns",
                    help="N>1: interleaved 8-pixel columns (every rank covers the whole picture: balanced) or contiguous row bands")
    ap.add_argument("--force-sharded", action="store_true", help="testing: run the N>1 code path (RCCL all-gather, comm stream) with a single rank")
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5],
                    help="BASELINE.json config: 2 = headline (Cornell-class), 4 = Sponza-class stand-in, 5 = glass stress stand-in")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist

    from hobbyrenderer_amd import native, scenes, structs as S

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rehearse = args.rehearse_on_one_gpu
    if rehearse:
        local_rank = 0
    sharded = world > 1 or args.force_sharded
    if sharded:
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    W, H, spp, bounces = args.width, args.height, args.spp, args.bounces
    luts = native.precompute_atmosphere()
    if args.config == 2:
        sc, view, pos, _ = scenes.config_cornell(luts, W, H)
        workload = "config 2: Cornell-box-class procedural scene (38 triangles, 9 instances, closed room, emissive quad + default sun)"
    elif args.config == 4:
        sc, view, pos, _ = scenes.config_sponza_class(luts, W, H)
        workload = "config 4 stand-in: Sponza-class procedural colonnade (~100k world triangles, textured PBR, MASK foliage, emissive, open sky)"
    else:
        sc, view, pos, _ = scenes.config_glass(luts, W, H)
        workload = "config 5 stand-in: glass stress (thick/thin dielectrics, Beer-Lambert, point + spot + sun) in the Cornell-class room"
    ctx = native.PathTracerContext(local_rank)
    ctx.upload_scene(sc)
    ctx.resize(W, H)
    lanes = [ctx]
    if args.frames_in_flight == 2 and not rehearse:
        ctx2 = native.PathTracerContext(local_rank)
        ctx2.upload_scene(sc)
        ctx2.resize(W, H)
        lanes.append(ctx2)
        for c in lanes:
            c.set_shadow_overlap(False)      # the other lane's frame fills the kernel tails; the intra-frame fork / join only costs then
    cb = scenes.fill_constants(view, pos, sc, 0, bounces)
    flags = {"default": S.FRAME_DEFAULT, "megakernel": S.FRAME_MEGAKERNEL, "wavefront": S.FRAME_WAVEFRONT}[args.mode]

    # row-band sharding (hobbyrenderer_amd/distributed.py); bands are contiguous in the row-major accumulation image
    from hobbyrenderer_amd.distributed import PipelinedFrames, band_for_rank, column_view, device_tensor, render_sharded
    y0, y1 = band_for_rank(H, world, rank)
    rows = y1 - y0
    accum_ptr, _ = ctx.device_images()
    full = device_tensor(accum_ptr, (H, W, 4), dev) if sharded else None

    def all_gather(full_t, band_t):
        if not rehearse:
            dist.all_gather_into_tensor(full_t, band_t)          # RCCL over xGMI
        else:                                                    # gloo through host memory (one-GPU rehearsal)
            host = torch.empty(full_t.shape, dtype=full_t.dtype)
            dist.all_gather_into_tensor(host, band_t.cpu())
            full_t.copy_(host)

    same_stream = sharded and not rehearse
    pipelined = same_stream and not args.serial_gather
    if not pipelined and sharded:
        lanes = lanes[:1]
        ctx.set_shadow_overlap(True)
    lane_streams = None
    if same_stream:
        # render, band clone, RCCL all-gather and resolve are all ordered on torch streams: no host sync in a step
        if pipelined and len(lanes) == 2:
            lane_streams = [torch.cuda.Stream(dev) for _ in lanes]
            for c, st_ in zip(lanes, lane_streams):
                c.set_stream(st_.cuda_stream)
        else:
            ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    # column interleaving needs the pipelined path (its gather re-assembles the image) and a width divisible by 8 * ranks
    columns = pipelined and args.sharding == "columns" and W % (8 * world) == 0
    stripes = (world, rank) if columns else (1, 0)

    def band_renderer(c):
        def render_band(b0, b1):
            c.render(cb, accum_count=spp, tile=(0, b0, W, b1), flags=flags, stripes=stripes)
            if not same_stream:
                c.synchronize()                                 # library stream -> host before the host-staged gather
        return render_band
    render_band = band_renderer(ctx)

    frames = None
    if pipelined:
        # frame k's all-gather + resolve run on a second stream while frame k+1 renders (hobbyrenderer_amd/distributed.py)
        images = [device_tensor(c.device_images()[0], (H, W, 4), dev) for c in lanes]
        views = [column_view(im, world, rank) if columns else im[y0:y1] for im in images]
        frames = PipelinedFrames([band_renderer(c) for c in lanes], views, H, W, rank, world, all_gather,
                                 lambda acc, out, stream: ctx.resolve_device(acc.data_ptr(), out.data_ptr(), H * W, stream), dev,
                                 lane_streams=lane_streams, layout="columns" if columns else "rows",
                                 # the step's product is the resolved image on every rank: shards -> Output in one pass, no assembled accumulation copy
                                 resolve_columns=lambda sh, acc, out, stream: ctx.resolve_columns_device(sh.data_ptr(), acc.data_ptr() if acc is not None else 0, out.data_ptr(), W, H, world, stream),
                                 keep_accumulation=False)
    step_no = [0]

    def step():
        if not sharded:
            lanes[step_no[0] % len(lanes)].render(cb, accum_count=spp, flags=flags)   # step k on lane k % L: L frames in flight
            step_no[0] += 1
        elif pipelined:
            frames.submit()
        else:
            render_sharded(render_band, full, rank, world, all_gather)                   # the single collective
            if not same_stream:
                torch.cuda.synchronize(dev)
            ctx.resolve_output()                                # Output = accum.rgb / accum.a on every rank

    def sync_all():
        if frames is not None:
            frames.finish()
        for c in lanes:
            c.synchronize()
        torch.cuda.synchronize(dev)
        if sharded:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # setup, not a step: every lane (context) renders once so that its queue pool is allocated and its kernels are loaded whatever W is
    for _ in range(len(lanes)):
        step()
    sync_all()
    for _ in range(args.warmup):
        step()
    sync_all()
    for c in lanes:
        c.reset_stats()
        c.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0

    class _Sum:                                               # ray counters of all lanes of this rank
        closestRays = sum(int(c.stats().closestRays) for c in lanes)
        shadowRays = sum(int(c.stats().shadowRays) for c in lanes)
        lastRenderMs = float(ctx.stats().lastRenderMs)
    st = _Sum
    if lane_streams is not None:
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    # reference point outside the timed region: the same steps one frame at a time (lane 0 only, frames back to back on one stream)
    one_at_a_time_ms = None
    if not sharded and len(lanes) > 1:
        n1 = max(2, min(args.steps, 10))
        ctx.synchronize()
        ctx.set_shadow_overlap(True)                         # what a one-frame-at-a-time host would use
        ctx.render(cb, accum_count=spp, flags=flags)
        ctx.synchronize()
        t1 = time.perf_counter()
        for _ in range(n1):
            ctx.render(cb, accum_count=spp, flags=flags)
        ctx.synchronize()
        one_at_a_time_ms = (time.perf_counter() - t1) / n1 * 1e3
        ctx.set_shadow_overlap(False)
    # per-kernel-class device times: a few extra steps with HRPT_FRAME_PROFILE (events around every launch), outside the timed region
    prof_steps = 0
    if args.mode != "megakernel":
        prof_steps = 3
        ctx.reset_stats()
        for _ in range(prof_steps):
            ctx.render(cb, accum_count=spp, tile=((0, 0, W, H) if columns else (0, y0, W, y1)) if sharded else (0, 0, 0, 0), flags=flags | S.FRAME_PROFILE, stripes=stripes)
        ctx.synchronize()
        pst = ctx.stats()

    red_dev = torch.device("cpu") if rehearse else dev
    tm = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    rays = torch.tensor([float(st.closestRays + st.shadowRays), float(st.closestRays), float(st.shadowRays)], dtype=torch.float64, device=red_dev)
    if sharded:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(tm.item())
    total_rays, closest_total, shadow_total = (float(x) for x in rays.tolist())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays / elapsed / 1e6
        result = {
            "metric": _baseline_metric(),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if not rehearse else "REHEARSAL on one GPU (gloo through host): not a result",
            "config": {"workload": f"{workload}, {W}x{H}, {spp} spp (accumulation indices 0..{spp - 1}), {bounces} bounces",
                       "sharding": (f"{world} rank(s), 8-pixel columns interleaved (column k on rank k mod {world}), BVH+scene replicated" if columns else f"{world} row band(s) of {rows} rows, BVH+scene replicated") + (", 1 RCCL all-gather of RGBA32F accumulation per step" + (" on a second stream, overlapped with the next step's render" if pipelined else "") if sharded else ""),
                       "frames_in_flight": len(lanes),
                       **({"ms_per_step_one_frame_in_flight": one_at_a_time_ms} if one_at_a_time_ms is not None else {}),
                       "mode": args.mode, "rays_per_step": total_rays / args.steps,
                       "closest_rays_per_step": closest_total / args.steps, "shadow_rays_per_step": shadow_total / args.steps},
        }
        # ---- cpu_baseline + n/t counters: oracle on a bounded sample (N=1 only; other N reuse the constants below)
        n_c = t_c = n_s = t_s = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle.binding import Oracle, OrStats
            o = Oracle(sc)
            ost = OrStats()
            cores = min(16, len(os.sched_getaffinity(0)))   # the CPU share of a one-GPU box
            sample_spp = spp                                 # the whole workload: ~10 s of CPU work on 16 cores
            t1 = time.perf_counter()
            o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, bounces), W, H, sample_spp, nthreads=cores, stats=ost)
            cpu_s = time.perf_counter() - t1
            o.close()
            d = ost.as_dict()
            n_c, t_c = d["closestNodes"] / max(1, d["closestRays"]), d["closestTris"] / max(1, d["closestRays"])
            n_s, t_s = d["shadowNodes"] / max(1, d["shadowRays"]), d["shadowTris"] / max(1, d["shadowRays"])
            result["cpu_baseline"] = {
                "value": (d["closestRays"] + d["shadowRays"]) / cpu_s / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": f"oracle (CPU restatement of the reference shader; the reference has no CPU path), same scene, {W}x{H}, "
                          f"accumulation indices 0..{sample_spp - 1} ({sample_spp} of {spp} spp), {bounces} bounces, {cores} pthreads, {cpu_s:.2f} s"}
        if n_c is None:
            # constants measured by the oracle on config 2 (Measurement); used when the oracle leg is skipped
            n_c, t_c, n_s, t_s = 16.14, 2.60, 17.37, 2.71
        b_closest = 768.0 + 32.0 * n_c + 48.0 * t_c
        b_shadow = 36.0 + 32.0 * n_s + 48.0 * t_s
        r0_closest, r0_shadow = float(st.closestRays) / args.steps, float(st.shadowRays) / args.steps   # rank 0, per step
        px0 = (W * H // world if columns else (y1 - y0) * W) if sharded else W * H
        step_bytes = r0_closest * b_closest + r0_shadow * b_shadow + px0 * spp * 48.0
        if prof_steps and pst.traceKernelLaunches > 0:
            # wavefront: per-class device time from HIP events the library records on ITS stream around every launch.
            # SURVEY 8(d) bytes split by the kernel that touches them (the three classes sum to B_closest / B_shadow):
            #   wf_extend: ray record 32 + traversal 32n+48t + hit record 20 per closest ray
            #   wf_shade : shading gather 588 + path state 128 per closest ray (+ 48 B/pixel/spp lives in raygen/resolve)
            #   wf_shadow: 36 + 32n+48t per shadow ray
            classes = {
                "wf_extend": (pst.traceKernelMs, pst.traceKernelLaunches, r0_closest * (52.0 + 32.0 * n_c + 48.0 * t_c)),
                "wf_shade": (pst.shadeKernelMs, pst.shadeKernelLaunches, r0_closest * (588.0 + 128.0)),
                "wf_shadow": (pst.shadowKernelMs, pst.shadowKernelLaunches, r0_shadow * b_shadow),
            }
            kernel = max(classes, key=lambda k: classes[k][0])
            tot_ms, n_launch, bytes_per_step = classes[kernel]
            launches = n_launch / prof_steps
            per_launch_bytes = bytes_per_step / launches
            avg_ms = tot_ms / n_launch
            kernel_times = {k: {"ms_per_step": v[0] / prof_steps, "launches_per_step": v[1] / prof_steps,
                                "algorithmic_GBps": v[2] / (v[0] / prof_steps * 1e-3) / 1e9} for k, v in classes.items()}
        else:
            # megakernel: one launch per accumulation index does the whole dispatch
            launches = spp
            per_launch_bytes = step_bytes / spp
            avg_ms = st.lastRenderMs / spp
            kernel = "pt_megakernel"
            kernel_times = None
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
        result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                              "note": "achieved = SURVEY 8(d) algorithmic bytes / measured launch time; a fraction above 1 means those bytes (BVH nodes, triangles, "
                                      "attribute records) are served from LDS and L2, not HBM: 'traffic' is the HBM traffic the PMC counters saw per launch, "
                                      "'valu' shows what actually bounds the kernel (VALU issue)",
                              "traffic": _pmc_traffic(kernel) if (args.config == 2 and world == 1 and (W, H, spp, bounces) == (1920, 1080, 8, 4)) else None,
                              "valu": _pmc_valu(kernel) if (args.config == 2 and world == 1 and (W, H, spp, bounces) == (1920, 1080, 8, 4)) else None, "kernel": kernel, "kernels": kernel_times, "avg_launch_ms": avg_ms, "launches_per_step": launches,
                              "algorithmic_bytes_per_launch": per_launch_bytes,
                              "bytes_per_closest_ray": b_closest, "bytes_per_shadow_ray": b_shadow,
                              "whole_step_algorithmic_GBps": step_bytes / (ms_per_step * 1e-3) / 1e9,
                              "n_closest": n_c, "t_closest": t_c, "n_shadow": n_s, "t_shadow": t_s}
        print(json.dumps(result))
    for c in lanes[1:]:
        c.close()
    ctx.close()
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
