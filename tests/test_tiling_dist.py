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
