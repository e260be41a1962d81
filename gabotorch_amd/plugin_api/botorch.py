"""`botorch` as the reference examples use it."""
import types as _types

try:                                     # (tests/test_real_package_branch_cpu.py)
    from botorch import acquisition, fit_gpytorch_model, models  # noqa: F401
except Exception:                        # noqa: BLE001
    from .. import models as _models

    class SingleTaskGP(_models.SingleTaskGP):
        """botorch.models.SingleTaskGP(train_X, train_Y, likelihood=None, covar_module=None) [3P] (examples/gabo_spd.py:174)"""

        def __init__(self, train_X, train_Y, likelihood=None, covar_module=None, **kwargs):
            if covar_module is None:
                raise ValueError("a covariance module is required (botorch's Matern default is not part of this package)")
            prior = getattr(likelihood, "noise_prior", None)
            con = getattr(likelihood, "noise_constraint", None)
            lower = float(getattr(con, "lower_bound", 1e-4)) if con is not None else 1e-4
            init = getattr(likelihood, "initial_value", None)
            super().__init__(train_X, train_Y, covar_module, noise_prior=prior, noise_lower_bound=lower,
                             initial_noise=None if init is None else float(init))
            self.likelihood = likelihood

    def fit_gpytorch_model(mll, **kwargs):
        """botorch.fit_gpytorch_model(mll) [3P]: L-BFGS-B on the marginal likelihood of mll.model (examples/gabo_spd.py:194)"""
        _models.fit_gpytorch_model(mll.model if hasattr(mll, "model") else mll, **kwargs)
        return mll

    class ExpectedImprovement(_models.ExpectedImprovement):
        def __init__(self, model, best_f, maximize=True, **kwargs):
            super().__init__(model, float(best_f), maximize=maximize)

    class PosteriorMean(_models.PosteriorMean):
        pass

    models = _types.SimpleNamespace(SingleTaskGP=SingleTaskGP)
    acquisition = _types.SimpleNamespace(ExpectedImprovement=ExpectedImprovement, PosteriorMean=PosteriorMean,
                                         analytic=_types.SimpleNamespace(ExpectedImprovement=ExpectedImprovement,
                                                                         PosteriorMean=PosteriorMean))
