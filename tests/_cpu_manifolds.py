"""TEST INFRASTRUCTURE: torch-CPU manifolds and kernel-mean costs (formulas of oracle/) used to run the
device-agnostic solver code on a machine without a GPU.  Never imported by the package."""
import math

import torch


class CpuSphere:
    def __init__(self, n):
        self._n, self._shape, self.dim, self.typicaldist = n, (n,), n - 1, math.pi
    inner = staticmethod(lambda x, u, v: (u * v).sum(-1))
    norm = staticmethod(lambda x, u: (u * u).sum(-1).sqrt())
    proj = staticmethod(lambda x, h: h - (x * h).sum(-1, keepdim=True) * x)
    egrad2rgrad = proj
    zerovec = staticmethod(torch.zeros_like)

    def ehess2rhess(self, x, eg, eh, u):
        return self.proj(x, eh) - (x * eg).sum(-1, keepdim=True) * u

    @staticmethod
    def retr(x, u):
        y = x + u
        return y / y.norm(dim=-1, keepdim=True)

    def transp(self, x1, x2, d):
        return self.proj(x2, d)

    def rand(self):
        import numpy as np
        x = np.random.randn(self._n)
        return x / np.linalg.norm(x)


def _sym(a):
    return 0.5 * (a + a.transpose(-1, -2))


class CpuSpd:
    def __init__(self, n):
        self._n, self.dim, self.typicaldist = n, n * (n + 1) // 2, math.sqrt(n * (n + 1) / 2)

    @staticmethod
    def inner(x, u, v):
        return (torch.linalg.solve(x, u) * torch.linalg.solve(x, v).transpose(-1, -2)).sum((-1, -2))

    def norm(self, x, u):
        return self.inner(x, u, u).clamp(min=0).sqrt()
    zerovec = staticmethod(torch.zeros_like)
    egrad2rgrad = staticmethod(lambda x, g: x @ _sym(g) @ x)
    ehess2rhess = staticmethod(lambda x, eg, eh, u: x @ _sym(eh) @ x + _sym(u @ _sym(eg) @ x))
    transp = staticmethod(lambda x1, x2, d: d)

    @staticmethod
    def retr(x, u):
        L = torch.linalg.cholesky(x)
        Li = torch.linalg.inv(L)
        lam, v = torch.linalg.eigh(_sym(Li @ u @ Li.transpose(-1, -2)))
        return L @ (v * torch.exp(lam).unsqueeze(-2)) @ v.transpose(-1, -2) @ L.transpose(-1, -2)


def sphere_kernel_mean_cost(Y, w, beta):
    """x: R x n  ->  -sum_j w_j exp(-beta acos(clamp <x, y_j>)^2)   (the cost of tests/golden/make_golden_tr.py)"""
    def cost(x):
        c = (x.double() @ Y.T).clamp(-1 + 1e-15, 1 - 1e-15)
        d = torch.acos(c)
        return -(w * torch.exp(-beta * d * d)).sum(-1)
    return cost


def spd_kernel_mean_cost(Y, w, beta):
    """X: R x d x d  ->  -sum_j w_j exp(-beta d_AI(X, Y_j)^2), fp64 throughout"""
    def cost(x):
        L = torch.linalg.cholesky(x.double())
        Li = torch.linalg.inv(L)
        m = Li.unsqueeze(1) @ Y.unsqueeze(0) @ Li.transpose(-1, -2).unsqueeze(1)
        lam = torch.linalg.eigvalsh(_sym(m))
        d2 = (torch.log(lam) ** 2).sum(-1) + 1e-15
        return -(w * torch.exp(-beta * d2)).sum(-1)
    return cost
