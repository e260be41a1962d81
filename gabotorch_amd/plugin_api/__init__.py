"""The third-party namespaces the reference's examples import, for machines where they are not installed (this image and the GPU
box have neither gpytorch, botorch nor pymanopt, and no network).

    from gabotorch_amd.plugin_api import gpytorch, botorch
    import gabotorch_amd.plugin_api.pymanopt.manifolds as pyman_man
    import gabotorch_amd.plugin_api.pymanopt.solvers as pyman_solvers

Each submodule re-exports the real package when it is importable and otherwise a namespace with exactly the names the reference
examples use (examples/bo_spd/benchmark_examples/gabo_spd.py:9-11, 165-203 and siblings), backed by gabotorch_amd.models /
gabotorch_amd._compat / gabotorch_amd.manifolds.  With that, an example of the reference runs after replacing the import prefixes
`BoManifolds.` -> `gabotorch_amd.` and `import gpytorch / botorch / pymanopt...` -> the lines above (tests/test_gpu_examples.py).
"""
from . import botorch, gpytorch, pymanopt  # noqa: F401
