"""select_rows (manifold_optimize.py of this package) picks the rows botorch's initialisation heuristics pick - models.initialize_q_batch /
initialize_q_batch_nonneg, the restatements of [3P] botorch.optim.initializers the sweep used through round 5 - draw for draw, for every
branch of the two heuristics (manifold_optimize.py:296-317 of the reference chooses between them)."""
import warnings

import numpy as np
import pytest
import torch

from gabotorch_amd import models
from gabotorch_amd.manifold_optimization.manifold_optimize import select_rows


def _reference(y, n, seed, nonneg, **kw):
    gen = torch.Generator()
    gen.manual_seed(seed)
    rows = torch.arange(y.shape[0]).reshape(-1, 1, 1)
    fn = models.initialize_q_batch_nonneg if nonneg else models.initialize_q_batch
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        picked = fn(X=rows, Y=torch.from_numpy(y), n=n, generator=gen, **kw)
    bad = any(issubclass(w.category, models.BadInitialCandidatesWarning) for w in caught)
    return picked.reshape(-1).numpy(), bad


CASES = {
    "ei_like": lambda rng: np.maximum(rng.standard_normal(2048), 0.0) * np.exp(rng.standard_normal(2048)),      # many zeros, a few large values
    "all_positive": lambda rng: rng.uniform(0.1, 1.0, 512),
    "few_positive": lambda rng: np.where(np.arange(300) < 5, 1.0 + np.arange(300), -1.0) * 1.0,
    "none_positive": lambda rng: -rng.uniform(0.1, 1.0, 256),
    "constant": lambda rng: np.full(128, 0.25),
    "tiny_alpha": lambda rng: np.concatenate([[1.0], np.full(400, 1e-9)]),                                     # alpha has to shrink
    "huge_spread": lambda rng: rng.standard_normal(700) * 400.0,                                                 # exp overflows: eta Z halved
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("nonneg", [True, False])
def test_select_rows_picks_the_rows_of_the_botorch_heuristics(name, nonneg):
    rng = np.random.default_rng(abs(hash(name)) % 1000)
    y = np.ascontiguousarray(CASES[name](rng), dtype=np.float64)
    for n in (1, 7, 64, y.shape[0]):
        for seed in (0, 12345):
            kw = {"eta": 2.0, "alpha": 1e-3} if (seed and nonneg) else ({"eta": 0.5} if seed else {})
            want, bad_want = _reference(y, n, seed, nonneg, **kw)
            gen = torch.Generator()
            gen.manual_seed(seed)
            got, bad = select_rows(y, n, gen, nonneg, kw.get("eta", 1.0), kw.get("alpha", 1e-4))
            assert got.dtype == np.int64 and got.shape == (n,)
            np.testing.assert_array_equal(got, want, err_msg=f"{name} n={n} seed={seed}")
            assert bad == bad_want


def test_select_rows_raises_on_nan_scores_instead_of_looping():
    """(the heuristics themselves never terminate on a NaN maximum: `while alpha_pos.sum() < n` with every comparison false)"""
    y = np.where(np.arange(200) == 17, np.nan, np.linspace(0.0, 1.0, 200))
    with pytest.raises(RuntimeError, match="NaN"):
        select_rows(y, 8, torch.Generator(), True)


def test_select_rows_refuses_more_restarts_than_samples():
    with pytest.raises(RuntimeError, match="cannot be larger"):
        select_rows(np.zeros(4), 5, torch.Generator(), True)
