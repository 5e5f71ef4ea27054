"""The inequality the second tree's closest-hit walks rest on (csrc/hrt_trace_packed.hpp, "Closest-hit walks over the SECOND tree"):
for a (ray, sphere) pair whose sphere test returns a valid t and whose own box test passes, the box GROWN as hrt_bvh.hip's
inflate_box grows a node has a computed slab entry <= t * (1 + 2^-7) -- so a node test at closest * (1 + 2^-7) never prunes an
instance that could still win.  Sampled in float32 numpy (the reference's IntersectAABB / IntersectSphere operation by operation)
on adversarial geometry: hits at the six poles (box faces) and on the silhouette, axis-parallel rays, rays that skim a face within
a few ulps of it, a huge ground sphere, coordinates over five decades.  The computed entry of the TIGHT box does exceed t in about
2 % of these candidates (the case the walker sends to the uploaded tree when it concerns the winner): asserted too, so that the
sample is known to reach the regime.  No GPU."""
import numpy as np
import pytest

f32 = np.float32


def _sample(n, seed, mode):
    rng = np.random.default_rng(seed)
    scale = 10.0 ** rng.uniform(-2, 3, n)
    c = (rng.uniform(-1, 1, (n, 3)) * scale[:, None]).astype(f32)
    r = (10.0 ** rng.uniform(-3, 0, n) * scale).astype(f32)
    if mode == "ground":
        c[:, 1] = -r; c[:, 0] *= f32(0.01); c[:, 2] *= f32(0.01)
    dist = (10.0 ** rng.uniform(-3, 2, n)) * r.astype(np.float64)
    dirn = rng.normal(size=(n, 3)); dirn /= np.linalg.norm(dirn, axis=1)[:, None]
    o64 = c.astype(np.float64) + dirn * (r.astype(np.float64) + dist)[:, None]
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    kind = rng.integers(0, 5, n)
    pole = np.eye(3)[rng.integers(0, 3, n)] * rng.choice([-1.0, 1.0], n)[:, None]
    u = np.where((kind == 1)[:, None], pole + 10.0 ** rng.uniform(-7, -2, n)[:, None] * rng.normal(size=(n, 3)), u)
    u /= np.linalg.norm(u, axis=1)[:, None]
    tgt = c.astype(np.float64) + u * r.astype(np.float64)[:, None] * np.where(kind == 2, 1.0 + 10.0 ** rng.uniform(-9, -3, n), 1.0)[:, None]
    d64 = tgt - o64; d64 /= np.linalg.norm(d64, axis=1)[:, None]
    if mode == "skim":          # origin within a few ulps of a face plane, ray nearly parallel to it
        ax = rng.integers(0, 3, n); sgn = rng.choice([-1.0, 1.0], n)
        o64 = c.astype(np.float64) + rng.uniform(-1.5, 1.5, (n, 3)) * r.astype(np.float64)[:, None]
        face = c[np.arange(n), ax].astype(np.float64) + sgn * r.astype(np.float64)
        o64[np.arange(n), ax] = face * (1.0 + rng.integers(-4, 5, n) * 2.0 ** -24)
        d64 = rng.normal(size=(n, 3)); d64[np.arange(n), ax] = -sgn * 10.0 ** rng.uniform(-8, -2, n)
        d64 /= np.linalg.norm(d64, axis=1)[:, None]
    if mode == "axis":
        ax = rng.integers(0, 3, n); d64 = np.eye(3)[ax] * np.sign(d64[np.arange(n), ax])[:, None]
        o64 = tgt - d64 * (dist + r)[:, None]
    return c, r, o64.astype(f32), d64.astype(f32)


def _check(c, r, o, d):
    with np.errstate(all="ignore"):
        inv = (f32(1) / np.where(d != 0, d, f32(1e-8))).astype(f32)                       # RTRay.cs:548-549
        lo = (c - r[:, None]).astype(f32); hi = (c + r[:, None]).astype(f32)              # Scene.cs:388-389

        def slab(lo, hi):                                                                 # SceneDeviceViews.cs:496-514
            t1 = ((lo - o) * inv).astype(f32); t2 = ((hi - o) * inv).astype(f32)
            tmn, tmx = np.minimum(t1, t2), np.maximum(t1, t2)
            return np.maximum(np.maximum(tmn[:, 0], tmn[:, 1]), tmn[:, 2]), np.minimum(np.minimum(tmx[:, 0], tmx[:, 1]), tmx[:, 2])
        tmin, tmax = slab(lo, hi)
        box = (tmax >= np.maximum(tmin, f32(0.001))) & (tmin <= f32(1e30))
        oc = (o - c).astype(f32)                                                          # SceneDeviceViews.cs:517-533
        a = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(f32); a = (a + d[:, 2] * d[:, 2]).astype(f32)
        b = (oc[:, 0] * d[:, 0] + oc[:, 1] * d[:, 1]).astype(f32); b = (b + oc[:, 2] * d[:, 2]).astype(f32); b = (f32(2) * b).astype(f32)
        cc = (oc[:, 0] * oc[:, 0] + oc[:, 1] * oc[:, 1]).astype(f32); cc = (cc + oc[:, 2] * oc[:, 2]).astype(f32); cc = (cc - r * r).astype(f32)
        disc = (b * b - (f32(4) * a * cc).astype(f32)).astype(f32)
        ok = disc >= 0
        sq = np.sqrt(np.where(ok, disc, 0)).astype(f32)
        ta = ((-b - sq) / (f32(2) * a)).astype(f32); tb = ((-b + sq) / (f32(2) * a)).astype(f32)
        t = np.where(ta < f32(0.001), tb, ta)
        hit = ok & ~(t < f32(0.001)) & (t > f32(0.001)) & (t < f32(1e29))
        # inflate_box (csrc/hrt_bvh.hip) on a node that is exactly this instance's box -- keep the two in step
        rho = (f32(0.5) * np.max(hi - lo, axis=1)).astype(f32)
        m = np.maximum(np.max(np.abs(lo), axis=1), np.max(np.abs(hi), axis=1)).astype(f32)
        s = (f32(1.0625) * (f32(4) * np.sqrt(f32(2) * rho * f32(2.0 ** -24) * m).astype(f32) + f32(2.0 ** -7) * rho + f32(2.0 ** -19) * m)).astype(f32) + f32(1e-30)
        lo2 = np.nextafter((lo - s[:, None]).astype(f32), f32(-np.inf)); hi2 = np.nextafter((hi + s[:, None]).astype(f32), f32(np.inf))
        grown, _ = slab(lo2, hi2)
    cand = box & hit
    return int(cand.sum()), int((cand & (tmin > t)).sum()), int((cand & ~(grown <= (t * f32(1.0 + 2.0 ** -7)).astype(f32))).sum())


@pytest.mark.parametrize("mode", ["free", "axis", "ground", "skim"])
def test_grown_box_is_entered_before_the_hit(mode):
    total = late = broken = 0
    for seed in range(6 if mode == "skim" else 3):
        n, a, b = _check(*_sample(400_000, 1000 + 17 * seed, mode))
        total += n; late += a; broken += b
    assert total > (10_000 if mode == "skim" else 500_000)          # skimming rays rarely pass both tests
    assert late > 0 or mode == "skim", "the sample never reaches the regime where the tight box is entered after the hit"
    assert broken == 0, "%d of %d candidates enter the grown box later than t (1 + 2^-7)" % (broken, total)
