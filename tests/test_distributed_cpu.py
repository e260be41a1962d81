"""world_size-2 `gloo` test of the restart sharding + all_gather/argmax of joint_optimize_manifold, on CPU with
the torch-CPU stand-in manifold and a toy non-negative acquisition."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    from tests._cpu_manifolds import CpuSphere
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((10, 3)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    Yt, w = torch.tensor(Y), torch.tensor(rng.uniform(0.2, 1.0, 10))

    def acq(X):                         # X: b x 1 x 3 -> b, non-negative
        c = (X[:, 0].double() @ Yt.T).clamp(-1 + 1e-15, 1 - 1e-15)
        d = torch.acos(c)
        return (w * torch.exp(-3.0 * d * d)).sum(-1)
    acq.is_nonnegative = True
    return acq, CpuSphere(3)


def _run(seed, num_restarts):
    from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
    from gabotorch_amd.manifold_optimization.manifold_optimize import joint_optimize_manifold
    acq, man = _problem()
    np.random.seed(seed)
    torch.manual_seed(seed)
    best = joint_optimize_manifold(acq, man, BatchedTrustRegions(), q=1, num_restarts=num_restarts, raw_samples=40, bounds=None)
    return best, acq(best[None])


def _worker(rank, world, port, num_restarts, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        best, val = _run(100 + rank, num_restarts)      # different seeds per rank: rank 0's initial conditions must win
        torch.save({"best": best, "val": val}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    for num_restarts in (5, 2):       # 5: uneven shards (3 + 2);  2: one restart per rank
        port = _free_port()
        mp.spawn(_worker, args=(2, port, num_restarts, str(tmp_path)), nprocs=2, join=True)
        r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
        r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
        assert torch.equal(r0["best"], r1["best"])                         # every rank returns the same candidate
        single, sval = _run(100, num_restarts)                              # same seed as rank 0 -> same initial conditions
        np.testing.assert_allclose(r0["best"].numpy(), single.numpy(), rtol=0, atol=1e-12)
        np.testing.assert_allclose(r0["val"].numpy(), sval.numpy(), rtol=1e-12)
        assert r0["best"].shape == (1, 3)


def test_shard_and_gather_helpers_single_process():
    from gabotorch_amd.manifold_optimization.manifold_optimize import gather_best, shard_restarts
    assert shard_restarts(7, 0, 3) == [0, 3, 6] and shard_restarts(7, 2, 3) == [2, 5] and shard_restarts(2, 3, 8) == []
    c = torch.arange(12.0).reshape(4, 1, 3)
    v = torch.tensor([0.1, 0.9, 0.9, 0.3])
    best, _, _ = gather_best(c, v, [0, 1, 2, 3], 4)
    assert torch.equal(best, c[1])                                          # ties -> lowest index


def _gram_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from gabotorch_amd.distributed import sharded_gram
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        x1, x2 = torch.randn(11, 4, generator=g, dtype=torch.float64), torch.randn(7, 4, generator=g, dtype=torch.float64)
        k = lambda a, b: torch.exp(-torch.cdist(a, b) ** 2)       # noqa: E731  stand-in kernel: the partition logic is under test
        b1, b2 = torch.randn(2, 3, 11, 4, generator=g, dtype=torch.float64), torch.randn(2, 3, 7, 4, generator=g, dtype=torch.float64)
        torch.save({"full": sharded_gram(k, x1, x2), "block": sharded_gram(k, x1, x2, gather=False), "ref": k(x1, x2),
                    "bfull": sharded_gram(k, b1, b2), "bblock": sharded_gram(k, b1, b2, gather=False), "bref": k(b1, b2)},
                   os.path.join(out_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_sharded_gram_row_blocks(tmp_path):
    from gabotorch_amd.distributed import row_block
    assert [row_block(11, r, 3) for r in range(3)] == [(0, 4), (4, 8), (8, 11)]
    assert [row_block(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    port = _free_port()
    mp.spawn(_gram_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"g{r}.pt")) for r in range(3)]
    for r, o in enumerate(outs):
        assert torch.equal(o["full"], o["ref"])
        lo, hi = row_block(11, r, 3)
        assert torch.equal(o["block"], o["ref"][lo:hi])
        # leading batch dimensions: only the rows (dim -2) are split
        assert o["bfull"].shape == (2, 3, 11, 7) and torch.equal(o["bfull"], o["bref"])
        assert torch.equal(o["bblock"], o["bref"][..., lo:hi, :])
