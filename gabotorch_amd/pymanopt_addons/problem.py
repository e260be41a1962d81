"""`Problem(manifold, cost, egrad=None, ehess=None, grad=None, hess=None, arg=None, precon=None, verbosity=2)` with the attributes the
reference's solvers read (BoManifolds/pymanopt_addons/problem.py:14-159): `manifold`, `cost(x) -> float`, `egrad(x)`, `grad(x)`,
`ehess(x, a)`, `hess(x, a)`, `precon(x, d)`, `verbosity`, and the assignable `_hess` slot (manifold_optimize.py:202).

Only the PyTorch route of the reference is restated (`arg=torch.Tensor()`, tools/autodiff/_pytorch.py:21-121): `cost` is a callable
on torch tensors returning a 0-dim tensor; points and tangent vectors cross this interface as numpy arrays (or lists of them on
product manifolds).  Value, gradient and Hessian-vector product at one point share one autograd graph, kept until a different point
arrives - pymanopt calls cost, grad and hess separately at the same x, and the torch tape needs the forward pass for each."""
import numpy as np
import torch


class _TorchEvaluator:
    def __init__(self, objective):
        self.objective = objective
        self._key = None          # the arrays of the last point
        self._x = None
        self._f = None
        self._df = None

    @staticmethod
    def _parts(x):
        seq = isinstance(x, (list, tuple))
        return (list(x) if seq else [x]), seq

    def _load(self, x):
        parts, seq = self._parts(x)
        arrs = [p.detach().cpu().numpy() if torch.is_tensor(p) else np.asarray(p) for p in parts]
        same = (self._key is not None and len(arrs) == len(self._key)
                and all(a.shape == k.shape and np.array_equal(a, k) for a, k in zip(arrs, self._key)))
        if not same:
            self._key = [a.copy() for a in arrs]
            self._x = [torch.from_numpy(a.copy()).requires_grad_(True) for a in arrs]
            self._f = self._df = None
        return seq

    def _value(self, seq):
        if self._f is None:
            f = self.objective(self._x if seq else self._x[0])
            if not torch.is_tensor(f) or f.dim() != 0:
                raise ValueError("the PyTorch route wants a cost function that returns a zero-dim tensor (a scalar)")
            self._f = f
        return self._f

    def _gradient(self, seq):
        if self._df is None:
            self._df = torch.autograd.grad(self._value(seq), self._x, create_graph=True, allow_unused=True)
        return self._df

    def cost(self, x):
        return self._value(self._load(x)).item()

    def egrad(self, x):
        seq = self._load(x)
        out = [(torch.zeros_like(xi) if d is None else d).detach().cpu().numpy() for d, xi in zip(self._gradient(seq), self._x)]
        return out if seq else out[0]

    def ehess(self, x, u):
        seq = self._load(x)
        us, useq = self._parts(u)
        if useq != seq or len(us) != len(self._x):
            raise ValueError("Incompatible lists in ehess")
        df = self._gradient(seq)
        r = sum((d.reshape(-1) * torch.as_tensor(np.asarray(ui)).to(d).reshape(-1)).sum() for d, ui in zip(df, us) if d is not None)
        if not (torch.is_tensor(r) and r.requires_grad):          # a gradient that does not depend on x (linear cost): zero Hessian
            out = [np.zeros_like(k) for k in self._key]
            return out if seq else out[0]
        h = torch.autograd.grad(r, self._x, retain_graph=True, allow_unused=True)
        out = [(torch.zeros_like(xi) if hi is None else hi).detach().cpu().numpy() for hi, xi in zip(h, self._x)]
        return out if seq else out[0]


class Problem:
    def __init__(self, manifold, cost, egrad=None, ehess=None, grad=None, hess=None, arg=None, precon=None, verbosity=2):
        self.manifold = manifold
        self._original_cost = cost
        self._cost = None
        self._egrad, self._ehess, self._grad, self._hess = egrad, ehess, grad, hess
        self._arg = arg
        self.precon = precon if precon is not None else (lambda x, d: d)
        self.verbosity = verbosity
        self._evaluator = None

    @property
    def backend(self):
        if self._evaluator is None:
            if not (callable(self._original_cost) and torch.is_tensor(self._arg) and self._arg.nelement() == 0):
                raise ValueError("Cannot determine autodiff backend: pass arg=torch.Tensor() with a cost on torch tensors "
                                 "(the only backend of the reference this package restates)")
            self._evaluator = _TorchEvaluator(self._original_cost)
        return self._evaluator

    @property
    def cost(self):
        if self._cost is None:
            self._cost = self.backend.cost
        return self._cost

    @property
    def egrad(self):
        if self._egrad is None:
            self._egrad = self.backend.egrad
        return self._egrad

    @property
    def grad(self):
        if self._grad is None:
            egrad = self.egrad
            self._grad = lambda x: self.manifold.egrad2rgrad(x, egrad(x))
        return self._grad

    @property
    def ehess(self):
        if self._ehess is None:
            self._ehess = self.backend.ehess
        return self._ehess

    @property
    def hess(self):
        if self._hess is None:
            ehess = self.ehess
            self._hess = lambda x, a: self.manifold.ehess2rhess(x, self.egrad(x), ehess(x, a), a)
        return self._hess
