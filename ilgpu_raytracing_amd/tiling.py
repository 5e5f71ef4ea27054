"""Row tiling of one frame over N GPUs, one process per GPU.

Every pixel of a frame is independent (threads write only their own index, the scene and
resPrev are read-only: RTRay.cs:141-142,203-325) and RNG / ReSTIR hashing key on the GLOBAL
pixel index (RTUtils.cs:108-113), so rank r simply renders rows [y0, y1) of the global image
with the full scene replicated on its GPU.  There is no data-path collective: each rank
copies its tile (hipMemcpy D2H inside hrt_render_frame) into ONE host framebuffer that all
ranks map from /dev/shm.  torch.distributed is used only for the barrier and the
max-over-ranks timing reduction.
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
