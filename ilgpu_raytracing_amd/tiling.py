"""Row tiling of one frame over N GPUs, one process per GPU.

Every pixel of a frame is independent (threads write only their own index, the scene and
resPrev are read-only: RTRay.cs:141-142,203-325) and RNG / ReSTIR hashing key on the GLOBAL
pixel index (RTUtils.cs:108-113), so rank r simply renders rows [y0, y1) of the global image
with the full scene replicated on its GPU.  Without ReSTIR reuse there is no data-path
collective: each rank copies its tile (hipMemcpy D2H inside hrt_render_frame) into ONE host
framebuffer that all ranks map from /dev/shm; torch.distributed is used only for the barrier
and the max-over-ranks timing reduction.

With reuse ON the path has one real exchange step (SURVEY.md 8e "caveat"): SpatialCompatible and
the temporal reprojection read the CURRENT G-buffer (worldPos, normalWS, objId; RTRay.cs:363-374)
and the PREVIOUS reservoirs (RTRay.cs:339-360, 488-515) of pixels anywhere in the image.
`render_reuse_frame` therefore runs a frame as launch 1 -> all-gather (28 B/pixel) -> launch 2 ->
all-gather of resCur (44 B/pixel), one collective per phase over all arrays of the phase, on the
device buffers themselves (RCCL when the process group is nccl; host staging for gloo rehearsals).
"""
import os

import numpy as np

from . import _types as T

ROW_GRANULE = 8   # a wave shades an 8x8 pixel tile: keep tile boundaries on multiples of 8


def partition_rows(height, world_size, rank):
    """Contiguous row block of `rank`: 8-row granules dealt as evenly as possible."""
    units = (height + ROW_GRANULE - 1) // ROW_GRANULE
    u0 = units * rank // world_size
    u1 = units * (rank + 1) // world_size
    return min(height, u0 * ROW_GRANULE), min(height, u1 * ROW_GRANULE)


class SharedFramebuffer:
    """Host framebuffer shared by all ranks of one node (numpy memmaps under /dev/shm).

    names: subset of the output names of hrt_outputs (T.OUTPUT_ARRAYS)."""

    def __init__(self, tag, width, height, names, create):
        self.width, self.height = width, height
        self.paths, self.arrays = {}, {}
        P = width * height
        base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        for n, dt, k in T.OUTPUT_ARRAYS:
            if n not in names:
                continue
            cnt = 1 if n == "cameraId" else P
            shape = (cnt, k) if k > 1 else (cnt,)
            path = os.path.join(base, "hrt_fb_%s_%s.bin" % (tag, n))
            self.paths[n] = path
            self.arrays[n] = np.lib.format.open_memmap(path, mode="w+" if create else "r+", dtype=dt, shape=shape) \
                if create else np.load(path, mmap_mode="r+")
        self._owner = create

    def outputs_struct(self):
        o = T.Outputs()
        for n, a in self.arrays.items():
            setattr(o, n, a.ctypes.data)
        return o

    def close(self, unlink=None):
        self.arrays = {}
        if unlink if unlink is not None else self._owner:
            for p in self.paths.values():
                try:
                    os.unlink(p)
                except OSError:
                    pass


# ------------------------------------------------------------------ inter-process tile exchange for ReSTIR reuse frames
def strip_rows(height, world_size, rank):
    """Rows of the 8-row strips rank owns (strips dealt round-robin, hrt_render_opts.strip_n / strip_i)."""
    rows = []
    for s in range(rank, (height + ROW_GRANULE - 1) // ROW_GRANULE, world_size):
        rows.extend(range(s * ROW_GRANULE, min(height, (s + 1) * ROW_GRANULE)))
    return np.asarray(rows, dtype=np.int64)


GBUFFER_EXCHANGE = (("gb_worldPos", np.float32, 3), ("gb_normalWS", np.float32, 3), ("gb_objId", np.int32, 1))
RESERVOIR_FIELDS = (("L", np.float32, 3), ("wi", np.float32, 3), ("pdf", np.float32, 1), ("w", np.float32, 1),
                    ("wSum", np.float32, 1), ("m", np.int32, 1), ("lightId", np.int32, 1))


class _DeviceArray:
    """Zero-copy view of library-owned device memory for torch (CUDA array interface v2)."""

    def __init__(self, ptr, shape, dtype):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": np.dtype(dtype).str, "data": (int(ptr), False), "version": 2}


def device_tensors(views, which, frame=0):
    """torch tensors [H, W*k] aliasing the device arrays of one exchange phase.
    which = "gbuffer" or "reservoir" (resCur of `frame`: set A on even frames, Framebuffer.cs:132-145)."""
    import torch
    h, w = views.height, views.width
    out = []
    if which == "gbuffer":
        for name, dt, k in GBUFFER_EXCHANGE:
            out.append(torch.as_tensor(_DeviceArray(getattr(views, name), (h, w * k), dt), device="cuda"))
    else:
        ptrs = views.res_a if (frame & 1) == 0 else views.res_b
        for i, (name, dt, k) in enumerate(RESERVOIR_FIELDS):
            out.append(torch.as_tensor(_DeviceArray(ptrs[i], (h, w * k), dt), device="cuda"))
    return out


def pack_rows(arrays, rows, pad_rows):
    """One flat uint8 buffer with rows `rows` of every [H, X] array (torch tensor or numpy), each padded to pad_rows rows."""
    import torch
    parts = []
    for a in arrays:
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(a)
        idx = torch.as_tensor(rows, device=t.device)
        sel = t.index_select(0, idx)
        if len(rows) < pad_rows:
            sel = torch.cat([sel, torch.zeros((pad_rows - len(rows), sel.shape[1]), dtype=sel.dtype, device=sel.device)], 0)
        parts.append(sel.contiguous().view(torch.uint8).reshape(-1))
    return torch.cat(parts)


def unpack_rows(arrays, rows, pad_rows, flat):
    """Inverse of pack_rows: writes rows `rows` of every array from the flat buffer of their owner."""
    import torch
    off = 0
    for a in arrays:
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(a)
        nbytes = pad_rows * t.shape[1] * t.element_size()
        block = flat[off:off + nbytes].view(t.dtype).reshape(pad_rows, t.shape[1])
        idx = torch.as_tensor(rows, device=t.device)
        t.index_copy_(0, idx, block[:len(rows)].to(t.device))
        off += nbytes


def all_gather_strips(arrays, height, world_size, rank, group=None):
    """All-gather of the strips every rank owns, over all `arrays` in ONE collective.  Device tensors go through the
    process group as they are when its backend is nccl (= RCCL, xGMI); any other backend stages through the host."""
    import torch
    import torch.distributed as dist
    rows = [strip_rows(height, world_size, r) for r in range(world_size)]
    pad = max(len(r) for r in rows)
    mine = pack_rows(arrays, rows[rank], pad)
    on_device = mine.is_cuda and dist.get_backend(group) == "nccl"
    send = mine if on_device or not mine.is_cuda else mine.cpu()
    recv = [torch.empty_like(send) for _ in range(world_size)]
    dist.all_gather(recv, send, group=group)
    for r in range(world_size):
        if r != rank:
            unpack_rows(arrays, rows[r], pad, recv[r])


def render_reuse_frame(renderer, params, world_size, rank, group=None, flags=0, outputs=None, tensors_of=None, sync=None):
    """One frame with ReSTIR reuse ON, tiled over the ranks of `group` (this rank: strips rank mod world_size).
    Returns (Stats of launch 1, Stats of launch 2).  Results are identical to a full-image render on one device.

    `renderer` needs render_params(params, outputs, flags=, strips=) with the flag semantics of hrt_render_frame.
    tensors_of(which, frame) -> the [H, X] arrays of one exchange phase ("gbuffer": GBUFFER_EXCHANGE, "reservoir":
    RESERVOIR_FIELDS of resCur); default: torch tensors aliasing the library's device arrays (hrt_device_buffers).
    sync(): waits for the exchange's device work; default torch.cuda.synchronize.  Both hooks exist so that the protocol
    itself can be driven on CPU ranks (tests/test_tiling_dist.py, gloo, the oracle as tile renderer)."""
    if tensors_of is None:
        import torch
        sync = sync or torch.cuda.synchronize
        tensors_of = lambda which, frame: device_tensors(renderer.device_views(), which, frame)      # noqa: E731
    sync = sync or (lambda: None)
    strips = (world_size, rank)
    st1 = renderer.render_params(params, None, flags=flags | T.FLAG_PRIMARY_ONLY, strips=strips)      # blocks until launch 1 is done
    all_gather_strips(tensors_of("gbuffer", params.frame), params.height, world_size, rank, group)
    sync()
    st2 = renderer.render_params(params, outputs, flags=flags | T.FLAG_SKIP_PRIMARY | T.FLAG_EXCHANGED, strips=strips)
    all_gather_strips(tensors_of("reservoir", params.frame), params.height, world_size, rank, group)
    sync()
    return st1, st2
