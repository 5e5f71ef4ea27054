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
def test_unary_bit_exact(orc, renderer, name):
    x = np.asarray(_cases()[name], dtype=np.float32)
    a = orc.math_eval(name, x)
    b = renderer.math_probe(orc.MATH_FN[name], x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("name", ["fmin", "fmax", "div", "atan2"])
def test_binary_bit_exact_including_specials(orc, renderer, name):
    rng = np.random.default_rng(12)
    spec = np.array([0.0, -0.0, 1.0, -1.0, np.nan, np.inf, -np.inf, 1e-40, -1e-40, 3.5, 1e38, -1e38], np.float32)
    X, Y = [g.reshape(-1) for g in np.meshgrid(spec, spec)]
    x = np.concatenate([X, rng.standard_normal(200000).astype(np.float32)])
    y = np.concatenate([Y, rng.standard_normal(200000).astype(np.float32)])
    a = orc.math_eval(name, x, y)
    b = renderer.math_probe(orc.MATH_FN[name], x, y)
    both_nan = np.isnan(a) & np.isnan(b)
    assert np.all((a.view(np.uint32) == b.view(np.uint32)) | both_nan)     # zero signs included


def test_device_code_is_not_contracted(renderer):
    rng = np.random.default_rng(13)
    x = (1 + rng.uniform(0, 1, 200000)).astype(np.float32)
    y = (1 + rng.uniform(0, 1, 200000)).astype(np.float32)
    got = renderer.math_probe(16, x, y)                                    # a*b + a
    want = (x * y).astype(np.float32) + x
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
