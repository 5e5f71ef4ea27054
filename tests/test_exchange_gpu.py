"""ReSTIR reuse frames tiled over several processes (SURVEY.md 8e caveat): launch 1 -> all-gather of the G-buffer ->
launch 2 -> all-gather of resCur, through HRT_FLAG_PRIMARY_ONLY / HRT_FLAG_EXCHANGED and the device pointers of
hrt_device_buffers.  One GPU is available here, so the ranks share it: first as two contexts in one process (the
exchange done by hand with the same pack/unpack helpers), then as two real processes over a gloo group
(tiling.render_reuse_frame; on a multi-GPU node the same code runs over nccl = RCCL with the tensors left on the device)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch          # before libhip_raytrace.so is loaded: torch brings its own HIP runtime of the same soname

from ilgpu_raytracing_amd import _types as T, engine, scenes, tiling
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(f, w, h, spp, prev_cam):
    cfg = scenes.CONFIGS[2]
    origin = (0.2 * f, 1.5, 5.5 - 0.15 * f)
    c2 = scenes.Config("mv", w, h, spp, origin, cfg.cam_lookat, extra=cfg.extra)
    return scenes.frame_params(c2, *H.host_funcs("hrt"), frame=f, reuse=True, prev_cam=prev_cam)


def test_partial_tile_reuse_needs_the_exchange_flags(renderer):
    s = engine.Scene(); scenes.build_config2(s); renderer.commit(s)
    p = _params(0, 64, 44, 1, None)
    with pytest.raises(Exception, match="exchange"):
        renderer.render_params(p, None, strips=(2, 0))
    with pytest.raises(Exception, match="exclude"):
        renderer.render_params(p, None, flags=T.FLAG_PRIMARY_ONLY | T.FLAG_SKIP_PRIMARY)
    renderer.render_params(p, None, flags=T.FLAG_PRIMARY_ONLY, strips=(2, 0))              # launch 1 alone is always allowed


@pytest.mark.parametrize("world", [2, 3])
def test_two_contexts_exchange_by_hand(orc, world):
    w, h, spp = 64, 44, 1
    rs = [engine.RTRenderer([0]) for _ in range(world)]
    full = engine.RTRenderer([0])
    try:
        for r in rs + [full]:
            s = engine.Scene(); scenes.build_config2(s); r.commit(s)
        rows = [tiling.strip_rows(h, world, k) for k in range(world)]
        pad = max(len(x) for x in rows)
        so = orc.OrcScene(); scenes.build_config2(so)
        A, B = H.new_reservoirs(w, h), H.new_reservoirs(w, h)
        prev_cam = None
        for f in range(3):
            p = _params(f, w, h, spp, prev_cam)
            for k, r in enumerate(rs):
                r.render_params(p, None, flags=T.FLAG_PRIMARY_ONLY, strips=(world, k))
            views = [r.device_views() for r in rs]
            gbs = [tiling.device_tensors(v, "gbuffer") for v in views]
            packed = [tiling.pack_rows(gbs[k], rows[k], pad) for k in range(world)]
            for k in range(world):
                for j in range(world):
                    if j != k:
                        tiling.unpack_rows(gbs[k], rows[j], pad, packed[j])
            torch.cuda.synchronize()
            outs = []
            for k, r in enumerate(rs):
                arrs, o = T.alloc_outputs(w, h)
                r.render_params(p, o, flags=T.FLAG_SKIP_PRIMARY | T.FLAG_EXCHANGED, strips=(world, k))
                outs.append(arrs)
            res = [tiling.device_tensors(v, "reservoir", f) for v in views]
            packed = [tiling.pack_rows(res[k], rows[k], pad) for k in range(world)]
            for k in range(world):
                for j in range(world):
                    if j != k:
                        tiling.unpack_rows(res[k], rows[j], pad, packed[j])
            torch.cuda.synchronize()
            # reference: the oracle on the full image
            ref, oo = T.alloc_outputs(w, h)
            prev, cur = (B, A) if f % 2 == 0 else (A, B)
            for n, a in cur.items():
                ref[n] = a; setattr(oo, n, a.ctypes.data)
            po = T.Outputs()
            for n, a in prev.items():
                setattr(po, n, a.ctypes.data)
            orc.render_frame(so.desc(), p, oo, po)
            for k in range(world):
                for n in ref:
                    if n == "cameraId":
                        continue
                    a = ref[n].reshape(h, -1)[rows[k]]
                    b = outs[k][n].reshape(h, -1)[rows[k]]
                    assert np.array_equal(a, b, equal_nan=True), (f, k, n)
            # and one full-image frame of a single context
            got, og = T.alloc_outputs(w, h)
            full.render_params(p, og)
            H.assert_outputs_equal(ref, got)
            prev_cam = engine.copy_camera(p.cam)
    finally:
        for r in rs + [full]:
            r.close()


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from ilgpu_raytracing_amd import _types as T, engine, scenes, tiling
from oracle import orc
from tests import helpers as H
from tests.test_exchange_gpu import _params
dist.init_process_group("gloo")                     # ranks share GPU 0 here; nccl needs one GPU per rank
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
w, h, spp = 96, 60, 2
r = engine.RTRenderer([0])
s = engine.Scene(); scenes.build_config2(s); r.commit(s)
tag = os.environ["HRT_TAG"]
names = ["color", "depth", "objectId", "radiance"] + H.RES_NAMES
if rank == 0:
    fb = tiling.SharedFramebuffer(tag, w, h, names, create=True)
dist.barrier()
if rank != 0:
    fb = tiling.SharedFramebuffer(tag, w, h, names, create=False)
so = orc.OrcScene(); scenes.build_config2(so)
A, B = H.new_reservoirs(w, h), H.new_reservoirs(w, h)
prev_cam = None
for f in range(3):
    p = _params(f, w, h, spp, prev_cam)
    tiling.render_reuse_frame(r, p, world, rank, outputs=fb.outputs_struct())
    dist.barrier()
    if rank == 0:
        ref, oo = T.alloc_outputs(w, h)
        prev, cur = (B, A) if f %% 2 == 0 else (A, B)
        for n, a in cur.items():
            ref[n] = a; setattr(oo, n, a.ctypes.data)
        po = T.Outputs()
        for n, a in prev.items():
            setattr(po, n, a.ctypes.data)
        orc.render_frame(so.desc(), p, oo, po)
        H.assert_outputs_equal(ref, fb.arrays, names=names)
    dist.barrier()
    prev_cam = engine.copy_camera(p.cam)
if rank == 0:
    print("REUSE_EXCHANGE_OK", world)
fb.close()
r.close()
dist.barrier()
dist.destroy_process_group()
'''


def test_two_processes_share_the_gpu_and_exchange_over_a_process_group(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, HRT_TAG="x%d" % os.getpid(), MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29300 + os.getpid() % 500), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "REUSE_EXCHANGE_OK 2" in out.stdout
