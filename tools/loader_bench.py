"""Throughput of the OBJ / MTL / TGA loader (host code): writes a synthetic OBJ of n x n quads with texcoords
and one 1024x1024 RLE TGA, then times hrth_mesh_load_obj (parse) and hrth_scene_load_obj_instance (parse + BVH build).
   python tools/loader_bench.py [--n 700]"""
import sys, os, time, json, argparse, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import engine
from tests import asset_kit as K

ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=700); a = ap.parse_args()
n = a.n
d = tempfile.mkdtemp(prefix="hrt_loader_")
K.write_tga(os.path.join(d, "big.tga"), K.checker(1024, 1024, 32, (200, 40, 40), (40, 200, 40)), image_type=10, depth=24)
with open(os.path.join(d, "m.mtl"), "w") as f:
    f.write("newmtl a\nKd 0.8 0.8 0.8\nmap_Kd big.tga\n")
xs = np.linspace(-10, 10, n + 1)
with open(os.path.join(d, "big.obj"), "w") as f:
    f.write("mtllib m.mtl\nusemtl a\n")
    for j in range(n + 1):
        f.write("".join("v %.6f %.6f %.6f\nvt %.5f %.5f\n" % (xs[i], 0.3 * np.sin(xs[i]) * np.cos(xs[j]), xs[j], i / n, j / n) for i in range(n + 1)))
    for j in range(n):
        base = j * (n + 1) + 1
        f.write("".join("f %d/%d %d/%d %d/%d %d/%d\n" % (base + i, base + i, base + i + 1, base + i + 1, base + i + n + 2, base + i + n + 2,
                                                         base + i + n + 1, base + i + n + 1) for i in range(n)))
size = os.path.getsize(os.path.join(d, "big.obj"))
t0 = time.perf_counter(); m = engine.load_obj(os.path.join(d, "big.obj"), 1.0, False); t1 = time.perf_counter()
s = engine.Scene(); t2 = time.perf_counter(); s.load_obj_instance(os.path.join(d, "big.obj")); t3 = time.perf_counter()
t4 = time.perf_counter(); img = engine.load_image(os.path.join(d, "big.tga")); t5 = time.perf_counter()
print(json.dumps({"obj_bytes": size, "triangles": int(len(m.triangles)), "parse_s": round(t1 - t0, 3), "parse_MBps": round(size / (t1 - t0) / 1e6, 1),
                  "load_obj_instance_s (parse + BLAS + TLAS)": round(t3 - t2, 3), "tga_1024x1024_rle_ms": round((t5 - t4) * 1e3, 2),
                  "note": "parse_s includes copying the arrays into numpy (engine.load_obj); one host thread"}))
for fn in os.listdir(d):
    os.unlink(os.path.join(d, fn))
os.rmdir(d)
