"""Phase statistics of the treelet walker (variant build -DHRT_TL_STATS):
   make -C ilgpu_raytracing_amd/csrc variant NAME=tlstats DEFS=-DHRT_TL_STATS
   HRT_LIB=ilgpu_raytracing_amd/csrc/variants/libhip_raytrace_tlstats.so python tools/tl_stats.py [config] [spp]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import _types as T, scenes, engine
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = engine.lib()
r = engine.RTRenderer([0])
s = engine.Scene(); scenes.build(cid, s); r.commit(s)
p = scenes.frame_params(scenes.CONFIGS[cid], engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=spp)
r.render_params(p, None, flags=T.FLAG_TREELETS)
buf = (C.c_ulonglong * 256)()
L.hrt_debug_tl_stats(buf)
st = r.render_params(p, None, flags=T.FLAG_TREELETS)
L.hrt_debug_tl_stats(buf)
a = np.array(list(buf), dtype=np.float64).reshape(8, 2, 16)
print("config %d spp %d: path stage %.2f ms" % (cid, spp, st.kernel_ms[1]))
for ph in range(8):
    for k, kn in ((0, "shadow"), (1, "closest")):
        v = a[ph, k]
        if v[13] == 0: continue
        it = max(v[3], 1)
        cyc = v[8] + v[9] + v[10] + v[11]
        print("slot %d %-7s waves %7d | rays %9d susp %9d done %9d | iter/wave %7.1f | lanes per iteration: lds-node %5.1f glob-node %5.1f leaf %5.1f idle %5.1f | rounds per iteration: lds %.2f glob %.2f"
              % (ph, kn, v[13], v[0], v[1], v[2], v[3] / v[13], v[4] / it / 64, v[5] / it / 64, v[6] / it / 64, v[7] / it / 64, v[14] / it, v[15] / it))
        print("        cycles per wave: refill %9.0f nodes %9.0f leaves %9.0f retire %9.0f staging %9.0f | per iteration: refill %6.0f nodes %6.0f leaves %6.0f retire %6.0f | share refill %.2f nodes %.2f leaves %.2f retire %.2f"
              % (v[8] / v[13], v[9] / v[13], v[10] / v[13], v[11] / v[13], v[12] / v[13], v[8] / it, v[9] / it, v[10] / it, v[11] / it, v[8] / cyc, v[9] / cyc, v[10] / cyc, v[11] / cyc))
