"""Image-tile sharding of the path-tracer pass over the GPUs of one node.

The reference is single-GPU. Pixels are independent units -- the RNG stream is a pure function of
(x, y, accumulation index) (src/shaders/RNG.hlsli:21-27) and accumulation is per pixel
(src/shaders/PathTracer.hlsl:332-339) -- so the image is split into contiguous row bands, the scene/BVH/LUTs
are replicated, every rank renders all accumulation indices of its band, and ONE all-gather of the RGBA32F
accumulation bands rebuilds the full image on every rank (RCCL over xGMI on GPUs; gloo in the CPU tests).
No other data-path collective exists.
"""


def band_for_rank(height, world, rank):
    """Contiguous row band [y0, y1) of `rank`; bands are equal so the all-gather shards are equal."""
    if height % world != 0:
        raise ValueError(f"image height {height} must be divisible by the number of ranks {world}")
    rows = height // world
    return rank * rows, (rank + 1) * rows


def render_sharded(render_band, accumulation, rank, world, all_gather):
    """One sharded frame.

    render_band(y0, y1): renders rows [y0, y1) of all accumulation indices into `accumulation` in place.
    accumulation: the full-image (H, W, 4) float32 tensor of this rank (torch tensor; device memory on GPUs).
    all_gather(full, band): collective writing every rank's band into `full` (torch.distributed.all_gather_into_tensor).
    Returns the band of this rank.
    """
    height = accumulation.shape[0]
    y0, y1 = band_for_rank(height, world, rank)
    render_band(y0, y1)
    if world > 1:
        band = accumulation[y0:y1].clone()      # 16 B/pixel * band; the only payload that crosses xGMI
        all_gather(accumulation, band)
    return y0, y1


def column_view(image, world, rank):
    """The 8-pixel columns k with k % world == rank of a (H, W, 4) image, as a strided view of shape (H, W / 8 / world, 8, 4).

    Column-interleaved sharding (hrpt_render stripeCount / stripeIndex): every rank works on all parts of the picture, so the
    cost differences between image regions (lamp, boxes, empty wall) average out; contiguous row bands of config 2 differ by 5-8 %."""
    h, w, _ = image.shape
    if w % (8 * world) != 0:
        raise ValueError(f"image width {w} must be a multiple of 8 * ranks = {8 * world}")
    return image.view(h, w // 8, 8, 4)[:, rank::world]


def columns_to_image(gathered, world, out):
    """Inverse of the rank-major all-gather of column_view shards: gathered (world * H * W / world * 4 floats, rank-major) -> out (H, W, 4)."""
    h, w, _ = out.shape
    c = w // 8 // world
    out.view(h, c, world, 8, 4).copy_(gathered.view(world, h, c, 8, 4).permute(1, 2, 0, 3, 4))
    return out


class _DevMem:
    """Exposes a library-owned device buffer through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, shape, device):
    """torch view of a float32 device buffer owned by libhobbyrt_pt.so (the accumulation / output images)."""
    import torch
    return torch.as_tensor(_DevMem(ptr, shape), device=device)


class PipelinedFrames:
    """Sharded frames with the all-gather of frame k overlapped with the render of frame k+1, and (optionally) two frames in flight.

    Per rank: a path-tracer context renders its row band into the context's accumulation image on a compute stream;
    the band is copied (device to device, 16 B/pixel) into one of two staging buffers; a second stream (`comm`) waits for
    that copy, runs the single RCCL all-gather into one of two full-size `gathered` images and resolves Output = rgb / a
    (hrpt_resolve_device) into the matching `output` image. A compute stream only waits for the comm stream when it is
    about to reuse a staging/gathered pair (two frames later), so a frame's xGMI traffic hides behind the next render.
    finish() joins all streams. On CPU (gloo tests) the same bookkeeping runs without streams.

    Lanes: `render_band` / `band_view` may be lists of L entries (one path-tracer context each) with `lane_streams` the L torch
    streams those contexts are bound to (hrpt_set_stream). Frame k is rendered by lane k % L, so L consecutive frames are in flight
    on the GPU at once: a band is too little work to fill 256 CUs through the tail of every kernel (an eighth of config 2, round 2:
    0.67 ms per frame with one lane, 0.52-0.55 with two, 0.48-0.51 with three, 0.47-0.50 with four, scripts/shard_host_overhead_probe.py). The gathers stay in frame order on the
    one comm stream (collectives must be issued in the same order on every rank).

    render_band(y0, y1): enqueues the band render on the lane's stream (the CURRENT torch stream when no lane_streams are given;
                         the context must be bound to it with hrpt_set_stream) and returns nothing.
    band_view: torch view of rows [y0, y1) of the context's accumulation image.
    resolve(accum_tensor, out_tensor, stream_handle): Output = rgb / a (hrpt_resolve_device on GPUs).
    """

    def __init__(self, render_band, band_view, height, width, rank, world, all_gather, resolve, device, lane_streams=None, layout="rows",
                 resolve_columns=None, keep_accumulation=True):
        import torch
        self.torch = torch
        self.render_band = list(render_band) if isinstance(render_band, (list, tuple)) else [render_band]
        self.band_view = list(band_view) if isinstance(band_view, (list, tuple)) else [band_view]
        if len(self.render_band) != len(self.band_view) or not 1 <= len(self.render_band) <= 4:
            raise ValueError("one to four lanes, each with a render_band and a band_view")
        self.lane_streams = list(lane_streams) if lane_streams is not None else None
        if self.lane_streams is not None and len(self.lane_streams) != len(self.render_band):
            raise ValueError("one stream per lane")
        self.all_gather, self.resolve = all_gather, resolve
        self.rank, self.world = rank, world
        # columns layout only: resolve_columns(shards, accumulation_or_None, output, stream_handle) fuses the re-assembly of the gathered
        # shards with the resolve (hrpt_resolve_columns_device); keep_accumulation=False then skips the assembled accumulation image
        self.resolve_columns, self.keep_accumulation = resolve_columns, keep_accumulation
        if layout not in ("rows", "columns"):
            raise ValueError("layout is 'rows' (contiguous row bands) or 'columns' (interleaved 8-pixel columns)")
        self.layout = layout
        self.gpu = device.type == "cuda"
        # staging / gathered / output slots: one per lane (at least two), frame k uses slot k % slots, so a frame's comm-stream work may still
        # be running when the frames of the other lanes are submitted
        self.slots = max(2, len(self.render_band))
        kw = dict(dtype=torch.float32, device=device)
        if layout == "rows":
            self.y0, self.y1 = band_for_rank(height, world, rank)
            self.staging = [torch.empty((self.y1 - self.y0, width, 4), **kw) for _ in range(self.slots)]
            self.gathered = [torch.empty((height, width, 4), **kw) for _ in range(self.slots)]
        else:
            # band_view is column_view(accumulation image of the lane, world, rank); render_band is called with the full row range
            if width % (8 * world) != 0:
                raise ValueError(f"image width {width} must be a multiple of 8 * ranks = {8 * world}")
            self.y0, self.y1 = 0, height
            self.staging = [torch.empty((height, width // 8 // world, 8, 4), **kw) for _ in range(self.slots)]
            self.shards = [torch.empty((world * height, width // 8 // world, 8, 4), **kw) for _ in range(self.slots)]   # rank-major concatenation, as the collective delivers
            self.gathered = [torch.empty((height, width, 4), **kw) for _ in range(self.slots)]
        self.output = [torch.empty((height, width, 4), **kw) for _ in range(self.slots)]
        self.frame = 0
        if self.gpu:
            self.comm = torch.cuda.Stream(device)
            self.rendered = [torch.cuda.Event() for _ in range(self.slots)]
            self.delivered = [torch.cuda.Event() for _ in range(self.slots)]
            # device time of the comm-stream work (all-gather + re-assembly + resolve) of the frame in each slot
            self.gather_begin = [torch.cuda.Event(enable_timing=True) for _ in range(self.slots)]
            self.gather_end = [torch.cuda.Event(enable_timing=True) for _ in range(self.slots)]
            # what a lane's stream actually waited for the comm stream (a slot coming round again before its gather + resolve were done)
            self.wait_pairs = []

    def submit(self):
        """Enqueue one frame; returns the slot (0/1) whose `gathered`/`output` images will hold it."""
        torch = self.torch
        s = self.frame % self.slots
        lane = self.frame % len(self.render_band)
        if self.gpu:
            main = self.lane_streams[lane] if self.lane_streams is not None else torch.cuda.current_stream()
            with torch.cuda.stream(main):
                self.render_band[lane](self.y0, self.y1)
                if self.frame >= self.slots:
                    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    w0.record(main)
                    main.wait_event(self.delivered[s])      # slot s is free again once the gather + resolve of the frame that used it are done
                    w1.record(main)
                    self.wait_pairs.append((w0, w1))
                self.staging[s].copy_(self.band_view[lane])  # the only payload that crosses xGMI
                self.rendered[s].record(main)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.rendered[s])
                self.gather_begin[s].record(self.comm)
                self._gather_and_resolve(s, self.comm.cuda_stream)
                self.gather_end[s].record(self.comm)
                self.delivered[s].record(self.comm)
        else:
            self.render_band[lane](self.y0, self.y1)
            self.staging[s].copy_(self.band_view[lane])
            self._gather_and_resolve(s, 0)
        self.frame += 1
        return s

    def _gather_and_resolve(self, s, stream_handle):
        if self.layout == "rows":
            self.all_gather(self.gathered[s], self.staging[s])
        else:
            self.all_gather(self.shards[s], self.staging[s])
            if self.resolve_columns is not None:                                   # one fused pass: shards -> (accumulation,) output
                self.resolve_columns(self.shards[s], self.gathered[s] if self.keep_accumulation else None, self.output[s], stream_handle)
                return
            columns_to_image(self.shards[s], self.world, self.gathered[s])     # one 16 B/pixel pass on the comm stream
        self.resolve(self.gathered[s], self.output[s], stream_handle)

    def gather_ms(self):
        """Device time of the last submitted frame's comm-stream work (all-gather + resolve); call after the streams were synchronised."""
        if not self.gpu or self.frame == 0:
            return 0.0
        s = (self.frame - 1) % self.slots
        return float(self.gather_begin[s].elapsed_time(self.gather_end[s]))

    def comm_wait_ms(self):
        """Total device time the lanes' streams spent waiting for the comm stream since the last call (after the streams were synchronised):
        0 when every all-gather + resolve hid behind the next renders."""
        if not self.gpu:
            return 0.0
        total = sum(float(a.elapsed_time(b)) for a, b in self.wait_pairs)
        self.wait_pairs = []
        return total

    def finish(self):
        """Make the current stream wait for every submitted frame (host synchronisation stays with the caller)."""
        if self.gpu:
            cur = self.torch.cuda.current_stream()
            cur.wait_stream(self.comm)
            for st in self.lane_streams or []:
                cur.wait_stream(st)
