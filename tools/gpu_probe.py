"""First-contact GPU probe: math bit-exactness, parity vs oracle on small frames, timing.
Run on the GPU box:  python tools/gpu_probe.py [--torch-first]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--torch-first" in sys.argv:
    import torch
    print("torch", torch.__version__, torch.cuda.is_available())
import numpy as np
from ilgpu_raytracing_amd import _types as T, scenes, engine
from oracle import orc

print(engine.lib().hrt_version(), "devices", engine.device_count())
r = engine.RTRenderer([0])

# ---- math exactness
rng = np.random.default_rng(1)
ok = True
tests = {
    "sin": rng.uniform(0, 6.2831855, 200000), "cos": rng.uniform(0, 6.2831855, 200000), "tan": rng.uniform(0.01, 1.5, 50000),
    "atan": rng.uniform(-50, 50, 50000), "acos": rng.uniform(-1, 1, 50000), "asin": rng.uniform(-1, 1, 50000),
    "rsqrt": np.abs(rng.standard_normal(200000)) * 10 ** rng.uniform(-20, 20, 200000), "sqrt": np.abs(rng.standard_normal(200000)) * 10 ** rng.uniform(-30, 30, 200000),
    "floor": rng.uniform(-1e5, 1e5, 50000), "round": np.concatenate([rng.uniform(-1e4, 1e4, 50000), np.arange(-100, 100) + 0.5]),
    "f2i": np.concatenate([rng.uniform(-3e9, 3e9, 50000), [np.nan, np.inf, -np.inf, 2147483648.0, -2147483648.0, 0.0, -0.0]]),
    "rcp": rng.standard_normal(200000) * 10 ** rng.uniform(-30, 30, 200000),
}
for name, x in tests.items():
    x = x.astype(np.float32)
    a = orc.math_eval(name, x); b = r.math_probe(orc.MATH_FN[name], x)
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    ok &= same
    print("math %-6s %s" % (name, "bit-exact" if same else "MISMATCH %d" % np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))))
spec = np.array([0.0, -0.0, 1.0, -1.0, np.nan, np.inf, -np.inf, 1e-40, -1e-40, 3.5], np.float32)
X, Y = [g.reshape(-1) for g in np.meshgrid(spec, spec)]
for name in ("fmin", "fmax", "div", "atan2"):
    xx = np.concatenate([X, rng.standard_normal(100000).astype(np.float32)]); yy = np.concatenate([Y, rng.standard_normal(100000).astype(np.float32)])
    a = orc.math_eval(name, xx, yy); b = r.math_probe(orc.MATH_FN[name], xx, yy)
    nan_both = np.isnan(a) & np.isnan(b)
    same = np.all((a.view(np.uint32) == b.view(np.uint32)) | nan_both)
    ok &= bool(same)
    print("math %-6s %s" % (name, "bit-exact" if same else "MISMATCH"))
    if not same:
        bad = np.nonzero(~((a.view(np.uint32) == b.view(np.uint32)) | nan_both))[0][:10]
        for i in bad: print("   ", xx[i], yy[i], a[i], b[i])
# contraction probe: fn 16 = a*b+a
xx = (1 + rng.uniform(0, 1, 100000)).astype(np.float32); yy = (1 + rng.uniform(0, 1, 100000)).astype(np.float32)
b = r.math_probe(16, xx, yy); a = (xx * yy).astype(np.float32) + xx
print("no-contraction:", np.array_equal(a.view(np.uint32), b.view(np.uint32)))

# ---- parity on small frames
def compare(cid, w, h, spp, builder=None, reuse_frames=0):
    cfg = scenes.CONFIGS.get(cid) or scenes.CONFIGS[1]
    so = orc.OrcScene(); sp = engine.Scene()
    if builder: builder(so); builder(sp)
    else: scenes.build(cid, so); scenes.build(cid, sp)
    r.commit(sp)
    p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, width=w, height=h, spp=spp)
    ao, oo = T.alloc_outputs(w, h); ag, og = T.alloc_outputs(w, h)
    t = time.time(); so_st = orc.render_frame(so.desc(), p, oo); t_cpu = time.time() - t
    st = r.render_params(p, og, flags=T.FLAG_COUNTERS)
    res = {}
    for k in ao:
        if k.startswith("res_"):
            continue
        a, b = ao[k], ag[k]
        if a.dtype == np.float32:
            eq = (a == b) | (np.isnan(a) & np.isnan(b))
        else:
            eq = a == b
        res[k] = int(np.count_nonzero(~eq))
    cnt_ok = all(so_st.k[i].as_dict() == st.k[i].as_dict() for i in range(2))
    rays = sum(st.k[i].rays_closest + st.k[i].rays_shadow for i in range(2))
    st2 = r.render_params(p, None)
    print("cfg%s %dx%d spp%d: mismatches %s counters_equal=%s | gpu ms prim %.3f path %.3f (%.1f Mrays/s) cpu %.2fs (%.2f Mrays/s)" % (
        cid, w, h, spp, {k: v for k, v in res.items() if v}, cnt_ok, st2.kernel_ms[0], st2.kernel_ms[1],
        rays / (st2.kernel_ms[0] + st2.kernel_ms[1]) / 1e3, t_cpu, rays / t_cpu / 1e6))
    if not cnt_ok:
        print("  oracle", so_st.k[1].as_dict()); print("  gpu   ", st.k[1].as_dict())
    return sum(res.values()) == 0 and cnt_ok

ok &= compare(1, 256, 256, 1)
ok &= compare(2, 480, 270, 4)
ok &= compare(3, 480, 270, 2)
ok &= compare("tex", 320, 240, 2, builder=scenes.build_textured_test_scene)
ok &= compare(4, 320, 180, 2, builder=lambda b: scenes.build_config4(b, 48, 48))
print("ALL OK" if ok else "FAILURES")

# ---- full-size timing config 2
cfg = scenes.CONFIGS[2]
sp = engine.Scene(); scenes.build(2, sp); r.commit(sp)
p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction)
st = r.render_params(p, None, flags=T.FLAG_COUNTERS)
rays = sum(st.k[i].rays_closest + st.k[i].rays_shadow for i in range(2))
for i in range(3):
    st = r.render_params(p, None)
    print("cfg2 1080p 4spp: prim %.3f ms path %.3f ms -> %.1f Mrays/s" % (st.kernel_ms[0], st.kernel_ms[1], rays / (st.kernel_ms[0] + st.kernel_ms[1]) / 1e3))
