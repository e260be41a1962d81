"""Counterpart of BoManifolds/pymanopt_addons: the `Problem` object the reference hands to its solvers."""
from .problem import Problem  # noqa: F401
