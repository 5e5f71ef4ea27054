"""include/hrt_math.h evaluated on gfx950 (hrt_math_probe) must return the oracle's bits: this is
what makes every data-dependent branch of the path tracer take the same side on both machines."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(11)
    wide = lambda n, lo, hi: (rng.standard_normal(n) * 10.0 ** rng.uniform(lo, hi, n)).astype(np.float32)
    return {
        "sin": rng.uniform(0, 6.2831855, 300000), "cos": rng.uniform(0, 6.2831855, 300000), "tan": rng.uniform(0.0, 1.55, 100000),
        "atan": rng.uniform(-1e3, 1e3, 100000), "acos": rng.uniform(-1, 1, 100000), "asin": rng.uniform(-1, 1, 100000),
        "rsqrt": np.abs(wide(300000, -30, 30)), "sqrt": np.abs(wide(300000, -38, 38)), "rcp": wide(300000, -38, 38),
        "floor": rng.uniform(-1e6, 1e6, 100000), "round": np.concatenate([rng.uniform(-1e4, 1e4, 100000), np.arange(-500, 500) + 0.5]),
        "f2i": np.concatenate([rng.uniform(-3e9, 3e9, 100000), [np.nan, np.inf, -np.inf, 2147483648.0, -2147483648.0, 2147483520.0, 0.0, -0.0]]),
    }


@pytest.mark.parametrize("name", list(_cases()))
def test_unary_bit_exact(orc, hooks_renderer, name):
    x = np.asarray(_cases()[name], dtype=np.float32)
    a = orc.math_eval(name, x)
    b = hooks_renderer.math_probe(orc.MATH_FN[name], x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("name", ["fmin", "fmax", "div", "atan2"])
def test_binary_bit_exact_including_specials(orc, hooks_renderer, name):
    rng = np.random.default_rng(12)
    spec = np.array([0.0, -0.0, 1.0, -1.0, np.nan, np.inf, -np.inf, 1e-40, -1e-40, 3.5, 1e38, -1e38], np.float32)
    X, Y = [g.reshape(-1) for g in np.meshgrid(spec, spec)]
    x = np.concatenate([X, rng.standard_normal(200000).astype(np.float32)])
    y = np.concatenate([Y, rng.standard_normal(200000).astype(np.float32)])
    a = orc.math_eval(name, x, y)
    b = hooks_renderer.math_probe(orc.MATH_FN[name], x, y)
    both_nan = np.isnan(a) & np.isnan(b)
    assert np.all((a.view(np.uint32) == b.view(np.uint32)) | both_nan)     # zero signs included


def test_device_code_is_not_contracted(hooks_renderer):
    rng = np.random.default_rng(13)
    x = (1 + rng.uniform(0, 1, 200000)).astype(np.float32)
    y = (1 + rng.uniform(0, 1, 200000)).astype(np.float32)
    got = hooks_renderer.math_probe(16, x, y)                                    # a*b + a
    want = (x * y).astype(np.float32) + x
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_fused_sincos_equals_sin_and_cos(orc, hooks_renderer):
    """hrt_sincos (device helper of the cosine-hemisphere sampler) returns the bits of hrt_sin and hrt_cos: every phi = 2*pi*k/2^24
    the sampler can produce from a 24-bit RNG draw on a stride, plus negative and large arguments."""
    k = np.arange(0, 1 << 24, 7, dtype=np.int64)
    phi = (np.float32(2.0) * np.float32(3.14159265358979323846)) * (k.astype(np.float32) * np.float32(1.0 / 16777216.0))
    rng = np.random.default_rng(14)
    x = np.concatenate([phi.astype(np.float32), rng.uniform(-50, 50, 200000).astype(np.float32), np.array([0.0, -0.0], np.float32)])
    s_ref, c_ref = orc.math_eval("sin", x), orc.math_eval("cos", x)
    s_got, c_got = hooks_renderer.math_probe(20, x), hooks_renderer.math_probe(21, x)
    assert np.array_equal(s_ref.view(np.uint32), s_got.view(np.uint32))
    assert np.array_equal(c_ref.view(np.uint32), c_got.view(np.uint32))


def test_sampler_variants_equal_the_general_functions(orc, hooks_renderer):
    """The cosine-hemisphere sampler calls variants that drop tests its arguments cannot need: hrt_sincos_nonneg (phi >= 0: no sign
    tests), sqrt_normal_range<NONZERO> (1 - r2 >= 2^-24: no zero select), rsqrt_clamped<FINITE> (|v|^2 < 100: no select for +inf).
    Each against the oracle's general function over every argument the sampler can produce / a wide sample of its domain."""
    k = np.arange(0, 1 << 24, dtype=np.float32) * np.float32(1.0 / 16777216.0)
    phi = (np.float32(2.0) * np.float32(3.14159265358979323846)) * k
    rng = np.random.default_rng(17)
    x = np.concatenate([phi, rng.uniform(0, 50, 200000).astype(np.float32)])
    assert np.array_equal(orc.math_eval("sin", x).view(np.uint32), hooks_renderer.math_probe(24, x).view(np.uint32))
    assert np.array_equal(orc.math_eval("cos", x).view(np.uint32), hooks_renderer.math_probe(25, x).view(np.uint32))
    one_minus = (np.float32(1.0) - k).astype(np.float32)
    wide = (10.0 ** rng.uniform(-20, 20, 2000000)).astype(np.float32)
    for v in (one_minus, wide):
        assert np.array_equal(orc.math_eval("sqrt", v).view(np.uint32), hooks_renderer.math_probe(26, v).view(np.uint32))
    fin = np.maximum((10.0 ** rng.uniform(-20, 38, 2000000)).astype(np.float32), np.float32(1e-20))
    assert np.array_equal(orc.math_eval("rsqrt", fin).view(np.uint32), hooks_renderer.math_probe(27, fin).view(np.uint32))


def test_sqrt_normal_range_equals_ieee_sqrt(orc, hooks_renderer):
    """The trimmed square root of the hemisphere sampler against the IEEE one: EVERY k / 2^24 (both sampler arguments r2 and
    1 - r2 are of that form) and a wide sample of its stated domain."""
    k = np.arange(0, 1 << 24, dtype=np.float32) * np.float32(1.0 / 16777216.0)
    rng = np.random.default_rng(15)
    wide = (10.0 ** rng.uniform(-20, 20, 2000000)).astype(np.float32)
    edge = np.array([0.0, 2.0 ** -96, 2.0 ** -24, 1.0, 1.0 - 2.0 ** -24, 4.0, 1e-20, 3.0e38], np.float32)
    for x in (k, np.float32(1.0) - k, wide, edge):
        a = orc.math_eval("sqrt", x)
        b = hooks_renderer.math_probe(22, x)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_trimmed_functions_equal_ieee_on_every_float_of_their_domain(orc, hooks_renderer):
    """rsqrt_clamped (the 1 / sqrt of every Normalize: x in [1e-20, +inf]) and sqrt_normal_range (+0, [2^-96, +inf]) against the
    IEEE definitions, exhaustively on the device: about 1.2 and 1.7 thousand million floats.  Plus a sample against the oracle's
    bits on x86-64, so that 'IEEE on the device' is itself pinned."""
    for which in (0, 1):
        bad, first = hooks_renderer.math_exhaustive(which)
        assert bad == 0, "function %d differs from IEEE for %d floats, first bits 0x%08X" % (which, bad, first)
    rng = np.random.default_rng(16)
    x = np.concatenate([(10.0 ** rng.uniform(-20, 38, 2000000)).astype(np.float32), np.array([1e-20, 1.0, 3.4e38, np.inf], np.float32)])
    x = np.maximum(x, np.float32(1e-20))
    a = orc.math_eval("rsqrt", x)
    b = hooks_renderer.math_probe(23, x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
