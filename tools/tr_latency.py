"""Latency anatomy of gabo_spd_tr_propose (first trust-region iteration of the config-4 sweep, all restarts active): time versus
the inner-iteration cap and versus the number of restarts."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gabotorch_amd import ops
from gabotorch_amd.fused_acquisition import FusedAcquisition
from gabotorch_amd.manifold_optimization import manifold_optimize as mo
from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch as to_vec, vector_to_symmetric_matrix_mandel_torch as to_mat
from tools.sweep_scaling import setup

dev = "cuda:0"
acq, man = setup(dev)
ops.set_error_checking(False)
np.random.seed(1); torch.manual_seed(1)
opts = {"device": dev, "batched_rand": True}
res = {}
for R in (64, 512, 2048, 4096, 8192):
    ic = mo.gen_batch_initial_conditions_manifold(acq, man, None, None, R, 4 * R, torch.float64, opts, to_vec)
    x = to_mat(ic[:, 0]).contiguous()
    fused = FusedAcquisition.build(acq, to_vec, torch.device(dev))
    fx, eg = fused.cost_egrad(x)
    g = man.egrad2rgrad(x, eg).contiguous()
    Delta = torch.full((R,), man.typicaldist / 8, dtype=torch.float64, device=dev)
    active = torch.ones(R, dtype=torch.uint8, device=dev)
    TR = ops.SpdTr(R, 5, 0, fused.acq_params(), fused.train.shape[0], dev)
    for mi in (1, 2, 4, 15):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            TR.propose(x, g, Delta, active, None, None, 0, 1e-6, 1.0, 0.1, 1, mi)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[f"R{R}_maxinner{mi}_us"] = round(min(ts) * 1e6, 1)
print(json.dumps(res, indent=1))
