"""`pymanopt` as the reference examples use it: `pymanopt.manifolds` and `pymanopt.solvers`."""
from . import manifolds, solvers  # noqa: F401
