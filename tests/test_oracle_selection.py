"""The oracle of the device-side restart selection (oracle/selection.py): its Philox4x32-10 against the known-answer vectors published with the
generator (Random123 kat_vectors: philox4x32, 10 rounds), and the heuristic's invariants."""
import numpy as np

from oracle import selection as osel


def test_philox4x32_10_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = osel.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert tuple(int(v) for v in got) == want


def test_selection_invariants():
    rng = np.random.default_rng(3)
    y = np.maximum(rng.standard_normal(2048), 0.0) * np.exp(rng.standard_normal(2048))
    for n in (1, 64, 512):
        picked, keys = osel.select_nonneg(y, n, seed=12345)
        assert picked.shape == (n,) and len(set(picked.tolist())) == n
        assert int(np.argmax(y)) in picked
        assert (keys[picked[:-1]] >= 0).all()
    u = osel.selection_uniforms(7, 100000)
    assert 0.0 < u.min() and u.max() <= 1.0 and abs(u.mean() - 0.5) < 5e-3
    assert osel.select_nonneg(-np.ones(8), 2, 1)[0] is None
    assert osel.select_nonneg(np.array([1.0, 0.0, 0.0, 0.0]), 2, 1)[0] is None
    # frequencies of a single draw follow the weights
    yy = np.array([1.0, 0.8, 0.5, 0.2])
    w = np.exp(2.0 * (yy / yy.max() - 1.0))
    counts = np.zeros(4)
    for seed in range(4000):
        counts[osel.select_nonneg(yy, 1, seed, eta=2.0, alpha=1e-9)[0][0]] += 1
    # (the arg-max is forced in when n = 1 misses it: a single draw always returns index 0 - so test the race itself through the keys)
    wins = np.zeros(4)
    for seed in range(4000):
        wins[int(np.argmax(osel.select_nonneg(yy, 1, seed, eta=2.0, alpha=1e-9)[1]))] += 1
    np.testing.assert_allclose(wins / 4000, w / w.sum(), atol=0.03)


def test_sphere_sampler_oracle_is_uniform_on_the_sphere():
    """oracle.selection.sphere_samples (what sphere_sample_kernel draws): unit rows, reproducible per (seed, index), independent of how many are drawn,
    and the distribution of [3P] Sphere.rand - coordinates with mean 0 and variance 1 / dim, no preferred direction."""
    a = osel.sphere_samples(99, 4096, 10)
    np.testing.assert_allclose(np.linalg.norm(a, axis=1), 1.0, rtol=0, atol=1e-15)
    np.testing.assert_array_equal(a[:100], osel.sphere_samples(99, 100, 10))          # addressed by sample index
    assert np.abs(a - osel.sphere_samples(100, 4096, 10)).max() > 0.1                  # another seed, another draw
    assert np.abs(a.mean(0)).max() < 0.02 and np.abs(a.var(0) - 0.1).max() < 0.01
    cov = a.T @ a / 4096
    assert np.abs(cov - np.eye(10) / 10).max() < 0.01
    odd = osel.sphere_samples(5, 1000, 7)                                              # odd dimension: the second deviate of the last pair is dropped
    np.testing.assert_allclose(np.linalg.norm(odd, axis=1), 1.0, rtol=0, atol=1e-15)
    assert odd.shape == (1000, 7) and np.abs(odd.mean(0)).max() < 0.06
