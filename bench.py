#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: Mrays/s at 1920x1080, 8 spp, 4 bounces (config 2: Cornell-box-class
procedural scene) on N MI355X of one node.

A "step" is one full render of that frame through the C ABI (hrpt_render with accumCount = spp) with the
Scene, BVH and LUTs already resident in HBM. N > 1: one process per GPU (torch.distributed over RCCL), the
image is sharded over the ranks (pixels are independent: RNG.hlsli:21-27), scene replicated, one all-gather of
the RGBA32F accumulation shards per step, then the resolve (Output = accum.rgb / accum.a). Total work is fixed
as N grows ("strong" scaling); value is the whole-job aggregate.

Rays = closest-hit queries + shadow queries actually launched (device counters; identical to the oracle's).

roofline (HBM, 8 TB/s): `achieved` = the bytes the dominant kernel class actually streams through its queues per launch
(records read / written x record size, from counters the kernels keep during the run: HrptStats::*QueueBytes) divided by
its average launch duration (HIP events the library records on ITS stream around every launch, in a separate
HRPT_FRAME_PROFILE pass of the same frame); `frac` = achieved / peak, always <= 1. `algorithmic_model` keeps SURVEY.md
8(d)'s per-ray figure (B_closest = 768 + 32 n + 48 t, B_shadow = 36 + 32 n + 48 t, + 48 B per pixel per spp, n / t counted
by the CPU oracle on this very config): it prices BVH-node, triangle and attribute touches as memory traffic although LDS /
L2 serve them, so it can exceed the peak and bounds nothing on small scenes. `traffic` (PMC FETCH_SIZE / WRITE_SIZE) and
`valu` (instruction-issue counters) come from committed rocprofv3 passes of this command and name their source file.
cpu_baseline: the oracle (a port; the reference has no CPU path) timed on all host cores of the box.
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


def _pmc_file(kind, config):
    """Newest committed rocprofv3 --pmc summary of `kind` ('traffic' | 'valu') for this config, or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{kind}_config{config}.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f), os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError):
        return None, None


def _host_cores():
    """(threads to use, cores visible): every core this process may run on, bounded by the cgroup CPU quota of the box when one is set
    (a one-GPU box exposes all 256 hardware threads of the host in its affinity mask but schedules a 16-core share: 256 oracle threads
    on that share measured 24 Mrays/s against 41 with 16)."""
    visible = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            if q != "max":
                quota = max(1, int(round(int(q) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = int(f.read()), int(g.read())
                if q > 0:
                    quota = max(1, int(round(q / period)))
        except (OSError, ValueError):
            pass
    return (min(visible, quota) if quota else visible), visible


def _baseline_metric():
    """The metric string exactly as BASELINE.json spells it (the file travels with the repository)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "Mrays/s at 1920\u00d71080, 8 spp, 4 bounces; 1/2/4/8 GPU scaling"


def _launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks as FRESH child processes through
    torch.distributed.run -- one per GPU, rendezvous on 127.0.0.1 at a free port -- relay their output (rank 0 prints the JSON line) and
    return their exit status. The parent imports neither torch nor the HIP library before this point (tests/test_bench_contract.py), and it
    starts children instead of replacing itself, so no process that has initialised a GPU is ever exec'ed over."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=None, help="default: BASELINE.json's value for the config (8 / 8 / 64 for configs 2 / 4 / 5)")
    ap.add_argument("--bounces", type=int, default=None, help="default: BASELINE.json's value for the config (4 / 8 / 12)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="threads of the cpu_baseline leg (default: every core this process may run on)")
    ap.add_argument("--mode", choices=["default", "megakernel", "wavefront"], default="default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal only: all ranks share GPU 0 and the all-gather runs over gloo through host memory; the printed value is NOT a result")
    ap.add_argument("--serial-gather", action="store_true", help="N>1: all-gather in stream order after each render instead of overlapping it with the next frame")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=(0, 1, 2, 3, 4),
                    help="independent steps (frames) in flight on each GPU: consecutive steps alternate between that many path-tracer contexts "
                         "(own streams, queues and images) so that one frame's kernel tails overlap the others'; 1 = one frame at a time; "
                         "0 = automatic: 2, or 3 when the image is sharded over several GPUs (a shard is too little work to fill the GPU through every "
                         "kernel's tail: one rank of eight, full pipeline: 0.56 ms per frame with 2 lanes, 0.51 with 3, 0.54 with 4)")
    ap.add_argument("--serial-kernels", action="store_true",
                    help="evidence runs under rocprofv3 --kernel-trace: one frame in flight and no shadow / extend overlap inside a frame, so every launch "
                         "runs alone and the tool's per-kernel averages are comparable with the HIP-event times of roofline.kernels (slower than the default)")
    ap.add_argument("--sharding", choices=("columns", "rows"), default="columns",
                    help="N>1: interleaved 8-pixel columns (every rank covers the whole picture: balanced) or contiguous row bands")
    ap.add_argument("--force-sharded", action="store_true", help="testing: run the N>1 code path (RCCL all-gather, comm stream) with a single rank")
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5],
                    help="BASELINE.json config: 2 = headline (Cornell-class), 4 = Sponza-class stand-in, 5 = glass stress stand-in")
    args = ap.parse_args(argv)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it has touched neither torch nor HIP) and exits with the ranks' status
        raise SystemExit(_launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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

    base_spp, base_bounces = {2: (8, 4), 4: (8, 8), 5: (64, 12)}[args.config]      # BASELINE.json configs[1] / [3] / [4]
    W, H = args.width, args.height
    spp = args.spp if args.spp is not None else base_spp
    bounces = args.bounces if args.bounces is not None else base_bounces
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
    n_lanes = args.frames_in_flight if args.frames_in_flight else (3 if world >= 2 else 2)
    if args.serial_kernels:
        n_lanes = 1
        ctx.set_shadow_overlap(False)
    if n_lanes > 1 and not rehearse:
        for _ in range(n_lanes - 1):
            c2 = native.PathTracerContext(local_rank)
            c2.upload_scene(sc)
            c2.resize(W, H)
            lanes.append(c2)
        for c in lanes:
            c.set_shadow_overlap(False)      # the other lanes' frames fill the kernel tails; the intra-frame fork / join only costs then
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
        if pipelined and len(lanes) >= 2:
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
    if frames is not None:
        frames.comm_wait_ms()                                   # drop the waits of the warm-up steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    comm_wait_ms = frames.comm_wait_ms() / max(1, args.steps) if frames is not None else 0.0

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
    my_elapsed = elapsed
    tm = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    rays = torch.tensor([float(st.closestRays + st.shadowRays), float(st.closestRays), float(st.shadowRays)], dtype=torch.float64, device=red_dev)
    if sharded:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(tm.item())
    total_rays, closest_total, shadow_total = (float(x) for x in rays.tolist())

    # per-rank breakdown for the sharded runs (the first hardware SCALE run should explain itself): render time of this rank's share
    # alone (HRPT_FRAME_PROFILE pass above, lastRenderMs = device time of one hrpt_render) and what the gather pipeline adds per step
    per_rank = None
    if sharded:
        mine = torch.tensor([float(pst.lastRenderMs) if prof_steps else 0.0, my_elapsed / args.steps * 1e3,
                             frames.gather_ms() if frames is not None else 0.0, comm_wait_ms], dtype=torch.float64, device=red_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        # render_ms_alone: device time of one hrpt_render of the rank's share with nothing else on the GPU; step_ms: the rank's own wall time per step in
        # the timed region; gather_resolve_ms: comm-stream work (all-gather + re-assembly + resolve) of one frame; comm_wait_ms_per_step: what the lanes'
        # streams waited for that work per step (0 = fully hidden behind the next renders)
        per_rank = [{"rank": i, "render_ms_alone": float(t[0]), "step_ms": float(t[1]), "gather_resolve_ms": float(t[2]), "comm_wait_ms_per_step": float(t[3])} for i, t in enumerate(allr)]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays / elapsed / 1e6
        result = {
            "metric": _baseline_metric(),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if not rehearse else "REHEARSAL on one GPU (gloo through host): not a result",
            "config": {"workload": f"{workload}, {W}x{H}, {spp} spp (accumulation indices 0..{spp - 1}), {bounces} bounces",
                       "baseline_config": args.config, "baseline_parameters": (spp, bounces) == (base_spp, base_bounces) and (W, H) == (1920, 1080),
                       "sharding": (f"{world} rank(s), 8-pixel columns interleaved (column k on rank k mod {world}), BVH+scene replicated" if columns else f"{world} row band(s) of {rows} rows, BVH+scene replicated") + (", 1 RCCL all-gather of RGBA32F accumulation per step" + (" on a second stream, overlapped with the next step's render" if pipelined else "") if sharded else ""),
                       "frames_in_flight": len(lanes), "serial_kernels": bool(args.serial_kernels),
                       "mode": args.mode, "rays_per_step": total_rays / args.steps,
                       "closest_rays_per_step": closest_total / args.steps, "shadow_rays_per_step": shadow_total / args.steps},
        }
        if one_at_a_time_ms is not None:
            # the reference drains the GPU every frame (src/Renderer.cpp:2053): the same steps with ONE frame in flight, a first-class figure
            result["one_frame_in_flight"] = {"ms_per_step": one_at_a_time_ms, "value": total_rays / args.steps / one_at_a_time_ms / 1e3, "unit": "Mrays/s"}
        if per_rank is not None:
            result["per_rank"] = per_rank
            steps_ms = [r["step_ms"] for r in per_rank]
            result["rank_balance"] = {"slowest_over_mean_step": max(steps_ms) / (sum(steps_ms) / len(steps_ms)),
                                      "slowest_over_mean_render_alone": (max(r["render_ms_alone"] for r in per_rank) / max(1e-9, sum(r["render_ms_alone"] for r in per_rank) / len(per_rank)))}
        # ---- cpu_baseline + n/t counters: the oracle on a bounded sample of THIS config (N = 1 only)
        model = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle.binding import Oracle, OrStats
            o = Oracle(sc)
            ost = OrStats()
            cores, visible = _host_cores()                              # every host core this process may use; both counts are reported
            cores = args.cpu_cores or cores
            # bounded sample: whole frames of the same workload, fewer accumulation indices when the full job would take minutes
            sample_spp = spp if args.config != 5 else min(spp, 8)
            t1 = time.perf_counter()
            o.render_accumulated(lambda i: scenes.fill_constants(view, pos, sc, i, bounces), W, H, sample_spp, nthreads=cores, stats=ost)
            cpu_s = time.perf_counter() - t1
            o.close()
            d = ost.as_dict()
            n_c, t_c = d["closestNodes"] / max(1, d["closestRays"]), d["closestTris"] / max(1, d["closestRays"])
            n_s, t_s = d["shadowNodes"] / max(1, d["shadowRays"]), d["shadowTris"] / max(1, d["shadowRays"])
            result["cpu_baseline"] = {
                "value": (d["closestRays"] + d["shadowRays"]) / cpu_s / 1e6, "unit": "Mrays/s", "cores": cores, "host_threads_visible": visible, "kind": "port",
                "sample": f"oracle (CPU restatement of the reference shader; the reference has no CPU path), same scene, {W}x{H}, "
                          f"accumulation indices 0..{sample_spp - 1} ({sample_spp} of {spp} spp), {bounces} bounces, {cores} pthreads "
                          f"(the box's CPU share: min(affinity mask = {visible}, cgroup quota)), {cpu_s:.2f} s"}
            b_closest = 768.0 + 32.0 * n_c + 48.0 * t_c
            b_shadow = 36.0 + 32.0 * n_s + 48.0 * t_s
            px0 = W * H
            step_bytes = closest_total / args.steps * b_closest + shadow_total / args.steps * b_shadow + px0 * spp * 48.0
            model = {"source": "SURVEY.md 8(d) byte model with n / t counted by the oracle on this config in this run; prices LDS- and L2-served "
                               "BVH / attribute touches as memory traffic, so it is an upper bound on useful traffic, not a roofline",
                     "n_closest": n_c, "t_closest": t_c, "n_shadow": n_s, "t_shadow": t_s,
                     "bytes_per_closest_ray": b_closest, "bytes_per_shadow_ray": b_shadow, "bytes_per_step": step_bytes,
                     "whole_step_GBps": step_bytes / (ms_per_step * 1e-3) / 1e9}
        if prof_steps and pst.traceKernelLaunches > 0:
            # wavefront: per-class device time (HIP events on the library's stream around every launch) and the queue bytes the class
            # moved in the same profiled steps (counters kept by the kernels; record sizes in pt_wavefront.hip, wavefront_queue_bytes)
            classes = {
                "wf_raygen": (pst.raygenKernelMs, pst.raygenKernelLaunches, pst.raygenQueueBytes),
                "wf_extend": (pst.traceKernelMs, pst.traceKernelLaunches, pst.traceQueueBytes),
                "wf_shade": (pst.shadeKernelMs, pst.shadeKernelLaunches, pst.shadeQueueBytes),
                "wf_shadow": (pst.shadowKernelMs, pst.shadowKernelLaunches, pst.shadowQueueBytes),
                "wf_resolve": (pst.resolveKernelMs, pst.resolveKernelLaunches, pst.resolveQueueBytes),
            }
            kernel = max(classes, key=lambda k: classes[k][0])
            tot_ms, n_launch, qbytes = classes[kernel]
            launches = n_launch / prof_steps
            per_launch_bytes = qbytes / max(1, n_launch)
            avg_ms = tot_ms / max(1, n_launch)
            kernel_times = {k: {"ms_per_step": v[0] / prof_steps, "launches_per_step": v[1] / prof_steps, "queue_bytes_per_step": v[2] / prof_steps,
                                "queue_GBps": (v[2] / (v[0] * 1e-3) / 1e9 if v[0] > 0 else None),
                                "frac_of_hbm_peak": (v[2] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if v[0] > 0 else None)} for k, v in classes.items()}
            all_ms = sum(v[0] for v in classes.values()); all_bytes = sum(v[2] for v in classes.values())
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            traffic_doc, traffic_src = _pmc_file("traffic", args.config)
            valu_doc, valu_src = _pmc_file("valu", args.config)
            at_baseline = world == 1 and (W, H, spp, bounces) == (1920, 1080, base_spp, base_bounces)
            # config 5's counters were collected at 8 of its 64 spp (per-launch figures do not depend on the spp count beyond the batch size)
            traffic = None
            # (config 5's PMC passes run at 8 of its 64 spp -- an eighth of the bytes per launch -- and list the three kernels of its shadow stage
            # separately: no per-launch figure comparable with `queue_bytes_per_launch` here; profiles/r02_pmc_config5.txt has the per-kernel bytes)
            if at_baseline and traffic_doc and args.config != 5:
                # launch-weighted over the instantiations of the class (the bounce-0 ones of a batch without raygen pass are listed as *_primary)
                rows = [v for k, v in traffic_doc.items() if k in (kernel, kernel + "_primary") and v.get("hbm_bytes_per_launch") is not None]
                n = sum(v.get("launches_sampled", 1) for v in rows)
                traffic = sum(v["hbm_bytes_per_launch"] * v.get("launches_sampled", 1) for v in rows) / n if n else None
            result["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "kernel": kernel, "avg_launch_ms": avg_ms, "launches_per_step": launches, "queue_bytes_per_launch": per_launch_bytes,
                "achieved_source": "in-run: HrptStats queue bytes (records read / written x record size, counted by the kernels) / HIP-event launch time",
                "traffic": traffic, "traffic_source": (traffic_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH doubled per the gfx950 "
                                                        "correction; committed file, not measured in this run)") if traffic is not None else None,
                "whole_step": {"queue_bytes": all_bytes / prof_steps, "kernel_ms": all_ms / prof_steps,
                               "queue_GBps": all_bytes / (all_ms * 1e-3) / 1e9, "frac": all_bytes / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "bytes_per_sample": all_bytes / prof_steps / (W * H * spp / world),
                               # the same bytes against the TIMED step (frames in flight overlap: the rate at which the design streams its queues
                               # through HBM while `value` is measured), this rank's share
                               "timed_region_GBps": all_bytes / prof_steps / (elapsed / args.steps) / 1e9,
                               "timed_region_frac": all_bytes / prof_steps / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
                "kernels": kernel_times,
                "valu": ({**valu_doc.get(kernel, {}),
                          # what the counters are read against: measured issue rates of a gfx950 SIMD per instruction class (cycles per wave64 instruction, >= 2 waves per SIMD)
                          "issue_rate_reference": {"full_rate_cycles": 2.2, "half_rate_cycles": 4.2, "quarter_rate_cycles": 8.2,
                                                   "full_rate": "v_fma/fmac/add/sub/mul_f32, v_and/or/xor, v_add/sub_u32, v_mov",
                                                   "half_rate": "v_min/max/min3/med3, v_cmp, v_cndmask, shifts, v_cvt, v_bfe/bfi/perm, 24-bit and 32-bit integer multiplies, DPP, v_readfirstlane, v_pk_*_f32",
                                                   "source": "profiles/r03_valu_issue_microbench.txt (scripts/microbench/valu_issue.hip)"},
                          "source": valu_src + " (committed rocprofv3 --pmc passes of this command, not measured in this run"
                          + ("; taken at 8 of the 64 spp, and for the kernel of that name only: the shadow stage's ray generation and any-hit pass have rows of their own in the file" if args.config == 5 else "") + ")"}
                         if (valu_doc and at_baseline) else None),
                "algorithmic_model": model,
                "queue_pool_bytes": int(pst.queuePoolBytes),
                "note": "frac = queue bytes the dominant kernel class (largest device time in this run) streams per launch / its launch time / 8 TB/s. "
                        + ("wf_shade is the kernel of the frame that streams: path records in, survivors and shadow entries out, and 'traffic' (PMC) next to "
                           "the counted bytes shows how little else it fetches. " if kernel == "wf_shade" else
                           "The traversal kernel is not HBM-bound: its BVH is LDS- or L2-resident and it is limited by VALU issue, request rate and lane "
                           "utilisation (see 'valu'). ")
                        + "On config 2 wf_extend and wf_shade take the same time to within 2 %, so which of the two is named here can change from run to "
                          "run: 'kernels' has time, bytes and frac_of_hbm_peak of every class, 'whole_step' the figure for all classes of a frame together.",
            }
        else:
            # megakernel: one launch per accumulation index does the whole dispatch; its HBM traffic is the 48 B per pixel of the image streams
            avg_ms = st.lastRenderMs / spp
            per_launch_bytes = W * H * 48.0 / world
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                  "kernel": "pt_megakernel", "avg_launch_ms": avg_ms, "launches_per_step": spp, "traffic": None,
                                  "algorithmic_model": model}
        print(json.dumps(result))
    for c in lanes[1:]:
        c.close()
    ctx.close()
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
