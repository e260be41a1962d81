"""The multi-rank branch of bench.py executed on ONE GPU: `python bench.py --gpus 2` starts its own two ranks through
torch.distributed.run; with GABO_BENCH_ONE_DEVICE set both ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one
device) - every other statement of the multi-rank path is the one the driver's 8-GPU run executes: self-launch, per-rank Gram builds,
row-block sharded Gram (left sharded / all_gathered), raw samples sharded by sample index + restarts sharded r % P with their two
all_gathers, the weak-scaling sweep, max-over-ranks timing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_two_ranks_on_one_device(tmp_path):
    env = dict(os.environ, GABO_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--preheat", "5",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1500)
    log = os.path.join(ROOT, "gpurun_out")
    os.makedirs(log, exist_ok=True)
    with open(os.path.join(log, "bench_two_ranks_one_device.log"), "w") as f:
        f.write(r.stdout + "\n--- stderr ---\n" + r.stderr[-20000:])
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["backend"] == "gloo" and line["world_size"] == 2 and line["collectives"]["ok"]
    assert line["value"] > 0 and line["parity"]["max_rel_err_vs_oracle_256x256"] < 1e-9
    sg = line["sharded_gram"]
    assert sg["left_sharded"]["ms"] > 0 and sg["all_gathered"]["ms"] >= sg["left_sharded"]["ms"] * 0.5
    assert sg["parity_max_abs_vs_unsharded"] < 1e-12              # the assembled Gram is the single-launch Gram
    sw = line["acq_sweep"]
    weak = sw["weak_scaling_512_restarts_per_gpu"]
    assert weak["restarts"] == 1024 and weak["restarts_per_s"] > 0
    assert "expected_scaling" in line and "cpu_baseline" not in line

    # the sweep whose raw samples come from the index-addressed device stream does not depend on the number of ranks: same
    # raw samples (shards of one stream), same selection, same restarts -> the best acquisition value of a single-rank run
    sys.path.insert(0, ROOT)
    from tools.sweep_bench import run_sweep
    single = run_sweep("cuda:0", num_restarts=512, device_rand=True, builtin_constraint=True)[2]
    assert abs(sw["best_acq_single_launch_solve_device_rand"] - single) <= 1e-9 * abs(single), (sw["best_acq_single_launch_solve_device_rand"], single)


def test_bench_one_rank_through_rccl(tmp_path):
    """The RCCL branch itself on hardware: bench.py under torch.distributed.run with ONE rank initialises the `nccl` backend on cuda:0 (device_id
    set) and executes every collective of the data path once - all_gather_into_tensor of the Gram slab, all_gather of the packed restart
    rows, all_reduce(MAX), barrier - before the timed region; the line names the backend it used."""
    import socket
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GABO_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--preheat", "5", "--no-cpu-baseline", "--no-sweep"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    log = os.path.join(ROOT, "gpurun_out")
    os.makedirs(log, exist_ok=True)
    with open(os.path.join(log, "bench_one_rank_rccl.log"), "w") as f:
        f.write(r.stdout + "\n--- stderr ---\n" + r.stderr[-20000:])
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["backend"] == "nccl" and line["world_size"] == 1 and line["n_gpus"] == 1
    assert line["collectives"]["ok"] and line["collectives"]["all_gather_into_tensor_gram_slab_ms"] > 0
    assert line["value"] > 0 and line["parity"]["max_rel_err_vs_oracle_256x256"] < 1e-9


def test_bench_eight_ranks_on_one_device(tmp_path):
    """The partition arithmetic of the driver's 8-GPU run, executed once: eight gloo ranks on cuda:0.  per = ceil(4096 / 8) = 512
    rows per rank in the sharded Gram with its padded all_gather slots, restarts r % 8 and raw samples by index range in the sweeps (512 restarts:
    64 per rank; 8192: 1024 per rank; weak: 4096), max-over-ranks timing, one JSON line."""
    env = dict(os.environ, GABO_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--preheat", "2",
                        "--no-cpu-baseline", "--no-symmetric", "--sweep-lite"], env=env, capture_output=True, text=True, timeout=1800)
    log = os.path.join(ROOT, "gpurun_out")
    os.makedirs(log, exist_ok=True)
    with open(os.path.join(log, "bench_eight_ranks_one_device.log"), "w") as f:
        f.write(r.stdout + "\n--- stderr ---\n" + r.stderr[-20000:])
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["world_size"] == 8 and line["backend"] == "gloo" and line["collectives"]["ok"]
    assert line["value"] > 0 and line["parity"]["max_rel_err_vs_oracle_256x256"] < 1e-9
    assert line["sharded_gram"]["parity_max_abs_vs_unsharded"] < 1e-12
    sw = line["acq_sweep"]
    assert sw["weak_scaling_512_restarts_per_gpu"]["restarts"] == 4096
    strong = sw["strong_scaling_fixed_total_restarts"]
    assert strong["512"]["n_gpus"] == 8 and strong["512"]["seconds"] > 0 and strong["8192"]["seconds"] > 0
    # the index-addressed device stream makes the sharded sweeps equal to the one-rank sweeps of the same total size
    sys.path.insert(0, ROOT)
    from tools.sweep_bench import run_sweep
    for total in (512, 8192):
        single = run_sweep("cuda:0", num_restarts=total, raw_samples=4 * total, device_rand=True, builtin_constraint=True)[2]
        got = strong[str(total)]["best_acq"]
        assert abs(got - single) <= 1e-9 * abs(single), (total, got, single)
    # every rank ran the native driver (round 6: it no longer steps aside when a process group exists), the 512-restart sweep with the selection
    # on the device; a rank's sweep of its 64 restarts costs what a one-rank 64-restart sweep costs, plus the two collectives (here: gloo through
    # the host, eight processes on one GPU - hence the slack)
    assert strong["512"]["native_sweep_on_every_rank"] and strong["512"]["selection_on_the_device_on_every_rank"]
    assert strong["8192"]["native_sweep_on_every_rank"]
    kw64 = dict(num_restarts=64, raw_samples=256, device_rand=True, builtin_constraint=True)
    run_sweep("cuda:0", **kw64)
    t64 = min(run_sweep("cuda:0", **kw64)[0] for _ in range(5))
    assert strong["512"]["seconds"] <= 2.0 * t64 + 1.5 * strong["512"]["seconds_of_the_two_collectives_alone"] + 1e-3, (strong["512"], t64)
