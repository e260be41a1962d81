"""Multi-GPU partitioning of the two data-parallel units of the path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

  * Gram build K(x1, x2): independent pairs.  x1/x2 are replicated (a few MB), rank r evaluates the row block
    [r*ceil(N1/P), ...) with the ordinary pairwise launch, and the blocks are either left sharded (when the consumer is
    sharded too) or assembled with ONE all_gather of N1/P x N2 doubles per rank.
  * Acquisition restarts: see manifold_optimization.manifold_optimize (interleaved restarts, all_gather + argmax).
"""
import torch


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def row_block(n_rows, rank, world):
    """[start, stop) of the row block owned by `rank` (equal ceil-sized blocks; the last ones may be short or empty)."""
    per = (n_rows + world - 1) // world
    start = min(rank * per, n_rows)
    return start, min(start + per, n_rows)


def sharded_gram(kernel_forward, x1, x2, gather=True):
    """K = kernel_forward(x1, x2) with the rows of x1 split over the ranks.  kernel_forward: any of this package's kernel
    `forward`s (or ops.spd_ai_pairwise / ops.sphere_pairwise partials).  Returns the full (N1, N2) matrix on every rank when
    `gather`, else this rank's (rows, N2) block.  x1 (..., N1, d) / x2 (..., N2, d) may carry leading batch dimensions (kept whole on
    every rank)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return kernel_forward(x1, x2)
    rank, world = dist.get_rank(), dist.get_world_size()
    n1 = x1.shape[-2]
    lo, hi = row_block(n1, rank, world)
    per = (n1 + world - 1) // world
    n2 = x2.shape[-2]
    lead = tuple(x1.shape[:-2])                     # leading batch dimensions stay whole on every rank: only the rows are split
    if hi > lo:
        block = kernel_forward(x1[..., lo:hi, :], x2)
    else:
        block = x1.new_zeros(lead + (0, n2), dtype=torch.float64)
    if not gather:
        return block
    # rows first for the collective: each rank contributes one contiguous (per, *lead, n2) slab
    pad = torch.zeros((per,) + lead + (n2,), dtype=torch.float64, device=block.device)
    pad[:hi - lo] = block.movedim(-2, 0)
    full = torch.empty((world * per,) + lead + (n2,), dtype=torch.float64, device=block.device)
    try:
        dist.all_gather_into_tensor(full, pad)           # one collective straight into the assembled matrix
    except (RuntimeError, NotImplementedError, AttributeError):
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        full = torch.cat(parts)
    return full[:n1].movedim(0, -2)
