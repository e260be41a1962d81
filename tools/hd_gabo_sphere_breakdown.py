import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/examples")
import numpy as np, torch
import hd_gabo_sphere as ex
import gabotorch_amd.manifold_optimization.manifold_gp_fit as mgf
import gabotorch_amd.nested_mappings.nested_spheres_optimization as nso
import gabotorch_amd.manifold_optimization.manifold_optimize as mo
T = {"fit": [], "recon": [], "sweep": []}
def wrap(mod, name, key):
    f = getattr(mod, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); T[key].append(time.perf_counter() - t); return r
    setattr(ex, name, g)
wrap(mgf, "fit_gpytorch_manifold", "fit"); wrap(nso, "optimize_reconstruction_parameters_nested_sphere", "recon"); wrap(mo, "joint_optimize_manifold", "sweep")
for dim, latent in ((5, 3), (21, 3), (51, 3)):
    for k in T: T[k].clear()
    t = time.perf_counter(); ex.run(dim, latent, 8, verbose=False); tot = time.perf_counter() - t
    print(dim, latent, {k: round(1e3 * float(np.median(v[1:])), 1) for k, v in T.items()}, "total/iter", round(1e3 * tot / 8, 1))
