"""N>1 path on CPU: two gloo ranks each render their row block and gather it into the shared host
framebuffer; rank 0 compares with a full-frame render.  The tile renderer here is the oracle (no GPU
in this container) -- what is under test is the product's partition / shared-framebuffer / barrier /
max-over-ranks plumbing that bench.py uses with the HIP renderer on the GPU box."""
import os
import subprocess
import sys

import numpy as np

from ilgpu_raytracing_amd import tiling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_rows_covers_image():
    for h in (1, 7, 8, 9, 270, 1080, 2160):
        for n in (1, 2, 3, 4, 8):
            parts = [tiling.partition_rows(h, n, r) for r in range(n)]
            assert parts[0][0] == 0 and parts[-1][1] == h
            for (a0, a1), (b0, b1) in zip(parts, parts[1:]):
                assert a1 == b0 and a0 <= a1
            for y0, y1 in parts[:-1]:
                assert y0 % 8 == 0 and (y1 % 8 == 0 or y1 == h)
    assert tiling.partition_rows(1080, 8, 3) == (400, 536)


WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from ilgpu_raytracing_amd import _types as T, scenes, tiling
from oracle import orc
from tests import helpers as H
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = scenes.CONFIGS[2]; w, h, spp = 96, 56, 2
names = ["color", "depth", "objectId", "radiance", "gb_worldPos", "gb_normalWS", "gb_baseColor", "gb_matId", "gb_objId", "gb_hitMask"]
tag = os.environ["HRT_TAG"]
if rank == 0:
    fb = tiling.SharedFramebuffer(tag, w, h, names, create=True)
dist.barrier()
if rank != 0:
    fb = tiling.SharedFramebuffer(tag, w, h, names, create=False)
so = orc.OrcScene(); scenes.build_config2(so)
p = scenes.frame_params(cfg, *H.host_funcs("orc", orc), width=w, height=h, spp=spp)
y0, y1 = tiling.partition_rows(h, world, rank)
dist.barrier(); t0 = time.perf_counter()
orc.render_frame(so.desc(), p, fb.outputs_struct(), row_begin=y0, row_end=y1, nthreads=1)
dist.barrier(); dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
dist.all_reduce(dt, op=dist.ReduceOp.MAX)
if rank == 0:
    ref, st, _ = H.oracle_frame(orc, scenes.build_config2, cfg, w, h, spp, nthreads=1)
    H.assert_outputs_equal(ref, fb.arrays, names=names)
    assert dt.item() > 0
    print("TILING_OK", world)
dist.barrier()
fb.close()
dist.destroy_process_group()
'''


def test_two_rank_row_tiling_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, HRT_TAG="t%d" % os.getpid(), MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29500 + os.getpid() % 500), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "TILING_OK 2" in out.stdout


def test_strip_rows_partition_the_image():
    for h in (1, 8, 9, 37, 270, 1080):
        for n in (1, 2, 3, 8):
            rows = np.concatenate([tiling.strip_rows(h, n, r) for r in range(n)])
            assert sorted(rows.tolist()) == list(range(h))
    assert tiling.strip_rows(37, 3, 1).tolist() == list(range(8, 16)) + list(range(32, 37))


def test_pack_unpack_rows_roundtrip():
    rng = np.random.RandomState(3)
    h, w = 37, 5
    src = [rng.rand(h, w * 3).astype(np.float32), rng.randint(-5, 5, (h, w)).astype(np.int32), rng.rand(h, w).astype(np.float32)]
    src[0][3, 4] = -0.0; src[0][5, 1] = np.nan                                 # bytes travel, not values
    dst = [np.zeros_like(a) for a in src]
    for r in range(3):
        rows = tiling.strip_rows(h, 3, r)
        tiling.unpack_rows(dst, rows, 16, tiling.pack_rows(src, rows, 16))
    assert all(a.tobytes() == b.tobytes() for a, b in zip(src, dst))


REUSE_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from ilgpu_raytracing_amd import _types as T, scenes, tiling
from oracle import orc
from tests import helpers as H
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = scenes.CONFIGS[2]; w, h, spp = 64, 44, 1            # 6 strips, the last one 4 rows: ragged tiles
so = orc.OrcScene(); scenes.build_config2(so)


class OracleTileRenderer:
    # what tiling.render_reuse_frame needs from a renderer, backed by the CPU oracle: whole-image host arrays of which only
    # this rank's strips are rendered, reservoir sets A / B picked by frame parity (Framebuffer.cs:132-145), and the
    # PRIMARY_ONLY / SKIP_PRIMARY flags of hrt_render_frame
    def __init__(self):
        self.arrs, self.o = T.alloc_outputs(w, h)
        self.A, self.B = H.new_reservoirs(w, h), H.new_reservoirs(w, h)

    def pair(self, frame):
        return (self.B, self.A) if frame %% 2 == 0 else (self.A, self.B)

    def render_params(self, p, outputs=None, flags=0, strips=None):
        prev, cur = self.pair(p.frame)
        for k, a in cur.items():
            self.arrs[k] = a; setattr(self.o, k, a.ctypes.data)
        po = T.Outputs()
        for k, a in prev.items():
            setattr(po, k, a.ctypes.data)
        assert bool(flags & T.FLAG_PRIMARY_ONLY) != bool(flags & T.FLAG_SKIP_PRIMARY)
        if flags & T.FLAG_SKIP_PRIMARY:
            assert flags & T.FLAG_EXCHANGED
        mode = 2 if (flags & T.FLAG_PRIMARY_ONLY) else False
        n, i = strips
        for s in range(i, (h + 7) // 8, n):
            orc.render_frame(so.desc(), p, self.o, po, row_begin=8 * s, row_end=min(h, 8 * s + 8), run_primary=mode, nthreads=1)
        return T.Stats()

    def tensors_of(self, which, frame):
        if which == "gbuffer":
            return [self.arrs[n].reshape(h, -1) for n, _, _ in tiling.GBUFFER_EXCHANGE]
        cur = self.pair(frame)[1]
        return [cur["res_" + n].reshape(h, -1) for n, _, _ in tiling.RESERVOIR_FIELDS]


tile = OracleTileRenderer()
full, fo = T.alloc_outputs(w, h)                           # rank 0 also renders the full image for comparison
FA, FB = H.new_reservoirs(w, h), H.new_reservoirs(w, h)
prev_cam = None
for f in range(3):
    origin = (0.2 * f, 1.5, 5.5 - 0.15 * f)                # moving camera: temporal reprojection crosses tiles
    c2 = scenes.Config("mv", w, h, spp, origin, cfg.cam_lookat, extra=cfg.extra)
    p = scenes.frame_params(c2, *H.host_funcs("orc", orc), frame=f, reuse=True, prev_cam=prev_cam)
    tiling.render_reuse_frame(tile, p, world, rank, tensors_of=tile.tensors_of)      # the product's protocol, unmodified
    arrs = tile.arrs
    if rank == 0:
        fprev, fcur = (FB, FA) if f %% 2 == 0 else (FA, FB)
        for k, a in fcur.items():
            full[k] = a; setattr(fo, k, a.ctypes.data)
        fpo = T.Outputs()
        for k, a in fprev.items():
            setattr(fpo, k, a.ctypes.data)
        orc.render_frame(so.desc(), p, fo, fpo, nthreads=1)
        mine = tiling.strip_rows(h, world, 0)
        for k in ("color", "radiance", "depth"):
            assert np.array_equal(arrs[k].reshape(h, -1)[mine], full[k].reshape(h, -1)[mine], equal_nan=True), (f, k)
        for k in H.RES_NAMES + ["gb_worldPos", "gb_normalWS", "gb_objId"]:
            assert arrs[k].tobytes() == full[k].tobytes(), (f, k)     # whole image after the exchange
    prev_cam = T.Camera.from_buffer_copy(bytes(p.cam))
if rank == 0:
    print("REUSE_TILING_OK", world)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_reuse_tile_exchange_gloo(tmp_path):
    """tiling.render_reuse_frame ITSELF (launch 1 -> all-gather G-buffer -> launch 2 -> all-gather resCur) on two gloo
    ranks, with the oracle behind the renderer interface it calls: 3 reuse frames with a moving camera end up
    byte-identical to full-image frames."""
    script = tmp_path / "reuse_worker.py"
    script.write_text(REUSE_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29000 + os.getpid() % 500), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "REUSE_TILING_OK 2" in out.stdout
