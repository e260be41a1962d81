"""TEST INFRASTRUCTURE: comparison of a BatchedTrustRegions trace with the reference traces of tests/golden/tr_traces.npz."""
import numpy as np


def compare_with_reference_trace(trace, g, prefix, atol_x, restarts=None):
    """trace: list of per-iteration dicts recorded by BatchedTrustRegions (generic path); g: the npz; prefix e.g. 'spd3_con_f64'.
    For every restart: walks the reference's outer iterations and checks radius (exact: Delta only ever changes by factors 2 and 1/4),
    tCG stop reason and iterate until the two runs part (a rounding-decided branch) or the reference ends.  Returns, per restart,
    (iterations in agreement, reference iterations, max |x - x_ref| over the agreeing iterations, the reference's radius at the
    iteration where the runs part or None, max |x_k - x_ref_k| over ALL common iterations - the drift after parting included)."""
    xs, delta, stop, nit = g[prefix + "_xs"], g[prefix + "_delta"], g[prefix + "_stop"], g[prefix + "_nit"]
    S = xs.shape[0]
    out = []
    for s in (range(S) if restarts is None else restarts):
        agree, worst, parted_at, drift = 0, 0.0, None, 0.0
        for k in range(min(int(nit[s]), len(trace), xs.shape[1] - 1)):
            t = trace[k]
            if not bool(t["active"][s]):
                break
            dx = float(np.max(np.abs(t["x"][s].cpu().numpy() - xs[s, k])))
            drift = max(drift, dx)
            if parted_at is not None:
                continue
            same = (float(t["Delta"][s]) == delta[s, k]) and (int(t["stop_inner"][s]) == int(stop[s, k])) and dx <= atol_x
            if not same:
                parted_at = float(delta[s, k])
                continue
            agree += 1
            worst = max(worst, dx)
        out.append((agree, int(nit[s]), worst, parted_at, drift))
    return out
