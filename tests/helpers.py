"""Shared helpers of the parity tests (oracle vs product)."""
import numpy as np

from ilgpu_raytracing_amd import _types as T, scenes


def bits_equal(a, b):
    """Element-wise LITERAL equality: floats are compared as their 32-bit patterns (so +0 != -0); the one tolerance
    is that any NaN equals any NaN (x86-64 and gfx950 propagate different NaN payloads).  Integers exact."""
    if a.dtype == np.float32:
        ua, ub = np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)
        return (ua == ub) | (np.isnan(a) & np.isnan(b))
    return a == b


def canon(a):
    """Copy of a (structured) array with every float NaN replaced by the one canonical quiet NaN, for byte comparisons of records
    computed on two machines: which NaN an operation returns (payload, sign) is the one thing x86-64 and gfx950 do not agree on."""
    a = np.array(a, copy=True)

    def fix(v):
        if v.dtype.names:
            for f in v.dtype.names:
                fix(v[f])
        elif v.dtype.kind == "f":
            v[np.isnan(v)] = np.float32(np.nan)
    fix(a)
    return a


def zero_sign_only(a, b):
    """Elements that are equal as floats but not as bit patterns (+0 vs -0): reported separately in failures."""
    if a.dtype != np.float32:
        return np.zeros(a.shape, dtype=bool)
    return (a == b) & ~bits_equal(a, b)


def assert_outputs_equal(ref, got, names=None, rows=None, width=None):
    bad = {}
    for k in ref:
        if names is not None and k not in names:
            continue
        a, b = ref[k], got[k]
        if rows is not None and k != "cameraId":
            a = a.reshape(-1, width, *a.shape[1:])[rows[0]:rows[1]]
            b = b.reshape(-1, width, *b.shape[1:])[rows[0]:rows[1]]
        n = int(np.count_nonzero(~bits_equal(a, b)))
        if n:
            bad[k] = (n, int(np.count_nonzero(zero_sign_only(a, b))))
    assert not bad, "arrays differ from the oracle {array: (elements, of which only in the sign of a zero)}: %s" % bad


def host_funcs(kind, orc=None):
    """(make_camera, bake, sun_dir) of the product host ('hrt') or of the oracle ('orc')."""
    if kind == "orc":
        return orc.camera_lookat, orc.camera_bake, orc.sun_dir
    from ilgpu_raytracing_amd import engine
    return engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction


def oracle_frame(orc, builder, cfg, w, h, spp, frame=0, reuse=False, prev=None, cur=None, rows=(0, 0), lock=0, prev_cam=None, nthreads=None):
    so = orc.OrcScene()
    builder(so)
    p = scenes.frame_params(cfg, *host_funcs("orc", orc), width=w, height=h, spp=spp, frame=frame, reuse=reuse,
                            rng_lock_noise=lock, prev_cam=prev_cam)
    arrs, o = T.alloc_outputs(w, h)
    if cur is not None:   # persistent reservoir arrays (ping-pong driven by the caller)
        for k, a in cur.items():
            arrs[k] = a
            setattr(o, k, a.ctypes.data)
    po = None
    if prev is not None:
        po = T.Outputs()
        for k, a in prev.items():
            setattr(po, k, a.ctypes.data)
    st = orc.render_frame(so.desc(), p, o, po, row_begin=rows[0], row_end=rows[1], nthreads=nthreads)
    return arrs, st, p


RES_NAMES = ["res_L", "res_wi", "res_pdf", "res_w", "res_wSum", "res_m", "res_lightId"]


def new_reservoirs(w, h):
    P = w * h
    out = {}
    for n, dt, k in T.OUTPUT_ARRAYS:
        if n in RES_NAMES:
            out[n] = np.zeros((P, k) if k > 1 else (P,), dtype=dt)
    return out


def algorithmic_bytes(k, n_pixels, launch):
    """ALGORITHMIC bytes of one launch from its work counters, using the
    reference's own struct sizes.  launch 0 = primary visibility, 1 = path trace."""
    d = k if isinstance(k, dict) else k.as_dict()
    fixed = 48 if launch == 0 else (64 + 12 + 44)
    return (n_pixels * fixed + d["node_visits"] * 44 + d["sphere_tests"] * (4 + 80) + d["tri_tests"] * (4 + 12 + 36)
            + d["tri_mt_hits"] * (4 + 44) + d["tri_accepted"] * (12 + 24) + d["leaf_instances"] * (4 + 144)
            + d["reuse_imports"] * (44 + 28))
