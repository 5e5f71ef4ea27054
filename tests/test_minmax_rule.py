"""Which Min / Max the kernels' arithmetic contract restates, and proof that the choice does not matter on BASELINE.

include/hrt_math.h: kernel code evaluates minNum / maxNum (a NaN operand is dropped: PTX min.f32, gfx950 v_min_f32); the
reference on ILGPU's CPUAccelerator -- north_star's parity target -- would evaluate .NET's Math.Min / Max (a NaN operand is
returned).  oracle/liborc_dotnet.so is the oracle compiled with the second rule (-DHRT_KERNEL_MINMAX_DOTNET).  Here:
  * every golden fixture and an 8-row strip of BASELINE configs 3 / 4 / 5 at full width render BYTE-EQUAL under both rules
    (config 3 at its 16 spp, configs 4 / 5 at 8 spp of 64 / 256 to keep the CPU suite short): no NaN reaches a min / max there;
  * the hostile scenes of tests/test_hostile_gpu.py are rendered under both and the frames that differ are COUNTED: that is the
    set on which "identical to the oracle" means "identical under the minNum rule" and says nothing about a CPUAccelerator."""
import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, scenes
from tests import helpers as H
from tests.golden import make_golden as G
from tests import test_hostile_gpu as HG


@pytest.fixture()
def both(orc):
    def run(fn):
        out = []
        for variant in ("", "dotnet"):
            orc.set_variant(variant)
            try:
                out.append(fn())
            finally:
                orc.set_variant("")
        return out
    return run


def _same(a, b):
    return all(np.ascontiguousarray(a[k]).tobytes() == np.ascontiguousarray(b[k]).tobytes() for k in a)


@pytest.mark.parametrize("name", list(G.CASES))
def test_golden_fixtures_do_not_depend_on_the_rule(orc, both, name):
    builder, cfg, w, h, spp, reuse_frames = G.CASES[name]

    def render():
        if reuse_frames == 0:
            return [H.oracle_frame(orc, builder, cfg, w, h, spp)[0]]
        res = [H.new_reservoirs(w, h), H.new_reservoirs(w, h)]
        frames = []
        for f in range(reuse_frames):
            cur, prev = res[f & 1], res[(f + 1) & 1]
            frames.append({k: v.copy() for k, v in H.oracle_frame(orc, builder, cfg, w, h, spp, frame=f, reuse=True, prev=prev, cur=cur)[0].items()})
        return frames
    a, b = both(render)
    assert all(_same(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("name,spp", [("full_config3_1080p_16spp_rows304", 16), ("full_config4_4k_64spp_rows1040", 8), ("full_config5_4k_256spp_rows800", 8)])
def test_baseline_strips_do_not_depend_on_the_rule(orc, both, name, spp):
    cfg_id, y0 = G.FULL_STRIPS[name]
    cfg = scenes.CONFIGS[cfg_id]
    builder = {3: scenes.build_config3, 4: scenes.build_config4, 5: scenes.build_config5}[cfg_id]

    def render():
        arrs, st, _ = H.oracle_frame(orc, builder, cfg, cfg.width, cfg.height, spp, rows=(y0, y0 + 8))
        return {k: arrs[k].reshape(cfg.height, cfg.width, *arrs[k].shape[1:])[y0:y0 + 8] for k in G.FULL_NAMES}, st.k[1].as_dict()
    (a, ca), (b, cb) = both(render)
    assert _same(a, b) and ca == cb
    assert np.all(np.isfinite(a["radiance"]))


def test_hostile_frames_that_depend_on_the_rule_are_counted(orc, both, capsys):
    differ = []
    for name, (builder, cfg, over) in HG.CASES.items():
        def render():
            so = orc.OrcScene(); builder(so)
            p = HG._frame(cfg, 96, 64, 2, over)("orc", orc)
            arrs, o = T.alloc_outputs(96, 64)
            orc.render_frame(so.desc(), p, o, None)
            return arrs
        a, b = both(render)
        if not all(bool(np.all(H.bits_equal(a[k], b[k]))) for k in a):
            differ.append(name)
    with capsys.disabled():
        print("\n[minmax rule] hostile frames whose picture depends on the Min / Max rule: %d of %d %s" % (len(differ), len(HG.CASES), differ))
    # the scenes WITHOUT a NaN in their inputs must not depend on it either
    assert not [n for n in differ if n not in ("nonfinite_spheres", "nonfinite_lights", "odd_transforms", "degenerate_spheres", "degenerate_mesh", "odd_textures")]
