"""The (c, -log c) table of the table-assisted fp64 log of the SPD Gram kernels (gabotorch_amd/csrc/gabo_log_tab.hpp) is the output of its
generator, and the routine built on it (modelled in numpy, same operation order as `log_tab` in gabo_device.hpp) is accurate to 2 ulp."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.sim import gen_log_table as glt  # noqa: E402


def test_committed_table_equals_generator_output():
    text = open(os.path.join(ROOT, "gabotorch_amd", "csrc", "gabo_log_tab.hpp")).read()
    body = text[text.index("{", text.index("kLogTab")) + 1:text.rindex("};")]
    vals = [float.fromhex(v) for v in re.findall(r"-?0x[0-9a-f.]+p[+-]?\d+", body)]
    assert len(vals) == 512
    c, l = glt.table()
    np.testing.assert_array_equal(np.array(vals[0::2]), c)
    np.testing.assert_array_equal(np.array(vals[1::2]), l)
    assert c[127] == 1.0 and l[127] == 0.0 and c[128] == 1.0 and l[128] == 0.0          # log is exact-in-form around 1


def test_table_log_model_accuracy():
    c, l = glt.table()
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-30, 30, 200000)), 1.0 + rng.uniform(-2e-2, 2e-2, 100000), 1.0 + rng.uniform(-1e-9, 1e-9, 20000),
                        np.array([1.0, 0.5, 2.0, 2.0 ** -0.5, 2.0 ** 0.5, 1e-300, 1e300, np.nextafter(1.0, 0.0), np.nextafter(1.0, 2.0)])])
    got, rmax = glt.log_tab_model(x, c, l)
    want = np.log(x.astype(np.longdouble))
    err = np.abs(got - want.astype(np.float64))
    ulp = np.spacing(np.abs(want.astype(np.float64)))
    assert rmax < 2.0 ** -7
    assert float(np.max(err / np.maximum(ulp, 5e-324))) <= 2.0
    assert glt.log_tab_model(np.array([1.0]), c, l)[0][0] == 0.0


def test_exp_table_equals_generator_output_and_model_accuracy():
    from tools.sim import gen_exp_table as get
    text = open(os.path.join(ROOT, "gabotorch_amd", "csrc", "gabo_exp_tab256.hpp")).read()
    body = text[text.index("{", text.index("kExp2Tab256")) + 1:text.rindex("};")]
    vals = np.array([float.fromhex(v) for v in re.findall(r"0x[0-9a-f.]+p[+-]?\d+", body)])
    tab = get.table()
    np.testing.assert_array_equal(vals, tab)
    assert tab[0] == 1.0 and abs(tab[128] - 2.0 ** 0.5) < 3e-16
    x = -np.random.default_rng(2).uniform(0, 40, 200000)
    got = get.exp_model(x, tab)
    want = np.exp(x.astype(np.longdouble)).astype(np.float64)
    assert float(np.max(np.abs(got - want) / want)) < 4.5e-16


def test_sphere_polynomial_equals_generator_output_and_epilogue_model():
    """asin(sqrt z)^2 / z of the sphere Gram epilogue (csrc/sphere_pairwise.hip): the committed coefficients are what
    tools/sim/fit_sphere_poly2.py produces, and a numpy model of the epilogue (cubic square root from a 2^-24 seed, sign indicator,
    magic-number rounding in the exp) stays within the conditioning of exp(-beta theta^2)."""
    import mpmath as mp
    sys.path.insert(0, os.path.join(ROOT, "tools", "sim"))
    import fit_sphere_poly2 as fsp
    text = open(os.path.join(ROOT, "gabotorch_amd", "csrc", "sphere_pairwise.hip")).read()
    body = text[text.index("#define GABO_SPH_P_COEFFS") + len("#define GABO_SPH_P_COEFFS"):text.index("__constant__ double kSphW")]
    vals = [float(v) for v in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", body)]
    coef = [float(c) for c in fsp.cheb_fit(fsp.p_true, mp.mpf(0), mp.mpf("0.5"), 17)]
    assert vals == coef
    z = np.linspace(0.0, 0.5, 2001)
    want = np.array([float(fsp.p_true(mp.mpf(float(v)))) for v in z])
    assert float(np.max(np.abs(fsp.horner64(coef, z) - want) / want)) < 5e-16
    rng = np.random.default_rng(3)
    c = np.concatenate([rng.uniform(-1, 1, 100000), 1 - 10.0 ** rng.uniform(-16, 0, 20000), -1 + 10.0 ** rng.uniform(-16, 0, 20000),
                        np.array([1.0, -1.0, 0.0, 1 + 2e-16, -1 - 2e-16])])
    for beta in (0.05, 1.2931471805599454, 40.0):
        got = fsp.epilogue_model(c, beta, coef, rng)
        want = np.exp(-beta * np.arccos(np.clip(c, -1.0 + 1e-15, 1.0 - 1e-15)) ** 2)
        assert float(np.max(np.abs(got - want) / want)) < 2.5e-14 * max(1.0, beta)
