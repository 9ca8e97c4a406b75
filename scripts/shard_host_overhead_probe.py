"""One rank's loop at N = 8 emulated on one GPU: PipelinedFrames (two lanes, columns layout of world 1, RCCL world 1) but every frame renders only
the 8-pixel columns k % 8 == 0, i.e. an eighth of the picture. Compares the steady-state time per frame with the bare render loop: the
difference is host submission cost + the comm-stream work of the real pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from hobbyrenderer_amd import native, scenes
from hobbyrenderer_amd.distributed import PipelinedFrames, column_view, device_tensor
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
luts = native.precompute_atmosphere()
W, H, N = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc, view, pos, cfg = scenes.config_cornell(luts, W, H)
cb = scenes.fill_constants(view, pos, sc, 0, 4)
lanes = []
for k in range(int(os.environ.get('LANES', '2'))):
    c = native.PathTracerContext(0); c.upload_scene(sc); c.resize(W, H); c.set_shadow_overlap(False); lanes.append(c)
streams = [torch.cuda.Stream(dev) for _ in lanes]
for c, st in zip(lanes, streams): c.set_stream(st.cuda_stream)
def bare(frames=200):
    for c in lanes: c.synchronize()
    t0 = time.perf_counter()
    for f in range(frames): lanes[f % len(lanes)].render(cb, accum_count=8, stripes=(N, 0))
    th = time.perf_counter() - t0
    for c in lanes: c.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3, th / frames * 1e3
views = [column_view(device_tensor(c.device_images()[0], (H, W, 4), dev), 1, 0) for c in lanes]
pf = None if len(lanes) < 2 else PipelinedFrames([(lambda c: (lambda a, b: c.render(cb, accum_count=8, stripes=(N, 0))))(c) for c in lanes], views, H, W, 0, 1,
                     lambda f, b: dist.all_gather_into_tensor(f, b), lambda acc, out, s: lanes[0].resolve_device(acc.data_ptr(), out.data_ptr(), H * W, s), dev,
                     lane_streams=streams, layout="columns",
                     **({} if os.environ.get("UNFUSED") else dict(keep_accumulation=False, resolve_columns=lambda sh, acc, out, s: lanes[0].resolve_columns_device(sh.data_ptr(), acc.data_ptr() if acc is not None else 0, out.data_ptr(), W, H, 1, s))))
def piped(frames=200):
    pf.finish(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for f in range(frames): pf.submit()
    th = time.perf_counter() - t0
    pf.finish(); torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / frames * 1e3, th / frames * 1e3
for rep in range(2):
    b = bare(); p = piped() if pf is not None else (0.0, 0.0)
    print(f"N={N}: bare render loop {b[0]:.3f} ms/frame (host submit {b[1]:.3f}); full pipeline {p[0]:.3f} ms/frame (host submit {p[1]:.3f})", flush=True)
dist.destroy_process_group()
