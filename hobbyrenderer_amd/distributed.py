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


class _DevMem:
    """Exposes a library-owned device buffer through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, shape, device):
    """torch view of a float32 device buffer owned by libhobbyrt_pt.so (the accumulation / output images)."""
    import torch
    return torch.as_tensor(_DevMem(ptr, shape), device=device)
