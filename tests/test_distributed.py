"""N>1 path on CPU: world_size-2 gloo processes exercise sharding + the episode-stats
all_gather used for reporting (the step() path itself has no collective)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from gym_auv_amd.distributed import shard_range


def test_shard_range_partitions_exactly():
    for n in (1, 7, 4096, 32768, 65537):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a0, a1), (b0, b1) in zip(blocks[:-1], blocks[1:]):
                assert a1 == b0
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gym_auv_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    lo, hi = D.shard_range(n_total, r, w)
    idx = torch.arange(lo, hi, dtype=torch.float32)
    stats = {"episode_return": idx * 2.0, "episode_length": idx + 100.0, "collision": (idx % 3 == 0).float()}
    full = D.gather_episode_stats(stats)
    t = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
    per_rank = D.gather_floats([10.0 * rank + 1.0, float(hi - lo)], torch.device("cpu"))   # (bench.py: per-rank world_gen_s, ...)
    assert D.backend_name() == "gloo"
    D.barrier()
    q.put((rank, lo, hi, {k: v.tolist() for k, v in full.items()}, t, per_rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_gloo_world2_gather_episode_stats():
    world, n_total = 2, 11                  # uneven shards: 6 + 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    exp = torch.arange(n_total, dtype=torch.float32)
    for rank, lo, hi, full, t, per_rank in res:
        assert per_rank == [[1.0, 6.0], [11.0, 5.0]]           # every rank sees every rank's figures, in rank order
        assert full["episode_return"] == (exp * 2).tolist()
        assert full["episode_length"] == (exp + 100).tolist()
        assert full["collision"] == (exp % 3 == 0).float().tolist()
        assert t == 2.0
    assert sorted((r[1], r[2]) for r in res) == [(0, 6), (6, 11)]


@pytest.mark.timeout(300)
def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts two ranks by itself; --dry-run
    shows their shards without touching a GPU.  Without enough GPUs (none here) and without --rehearse it
    must fail loudly instead of running one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--envs", "100", "--dry-run", "1"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr
    rows = sorted((json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")), key=lambda r: r["rank"])
    assert [r["rank"] for r in rows] == [0, 1] and all(r["world"] == 2 for r in rows)
    assert [(r["env_lo"], r["env_hi"]) for r in rows] == [shard_range(200, 0, 2), shard_range(200, 1, 2)] == [(0, 100), (100, 200)]
    assert [(r["seed_lo"], r["seed_hi"]) for r in rows] == [(1000, 1100), (1100, 1200)]
    if torch.cuda.device_count() < 2:
        bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                             env=env, capture_output=True, text=True, timeout=280)
        assert bad.returncode != 0 and "GPU(s) visible" in (bad.stderr + bad.stdout)
        # a launcher-provided WORLD_SIZE larger than the device count is refused too
        bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                             env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"), capture_output=True, text=True, timeout=280)
        assert bad.returncode != 0 and "GPU(s) visible" in (bad.stderr + bad.stdout)


def test_bench_world_seeds_do_not_depend_on_the_shard():
    """env g meets the world of seed 1000 + k * total + g in its k-th episode, however the batch is sharded."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    total, wpe = 24, 3
    want = {(g, k): 1000 + k * total + g for g in range(total) for k in range(wpe)}
    for world in (1, 2, 3):
        n_local = total // world
        got = {}
        for rank in range(world):
            lo, hi = shard_range(total, rank, world)
            seeds = bench.world_seeds(lo, n_local, total, wpe)
            assert len(seeds) == n_local * wpe
            for w, sd in enumerate(seeds):
                # auto-reset walks w -> (w + n_local) % W: local env e sees worlds e, e + n_local, ...
                got[(lo + w % n_local, w // n_local)] = int(sd)
        assert got == want


@pytest.mark.timeout(300)
def test_bench_under_an_outside_launcher_with_more_ranks_than_gpus():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --rehearse 1` -- the driver's launcher with
    more ranks than visible GPUs (round 2's only launcher-driven attempt failed right here with "invalid device
    ordinal"; the fix was never run): every rank gets through argument handling, sharding and device selection
    (device = LOCAL_RANK modulo the visible GPUs, gloo instead of RCCL) and prints its shard (--dry-run)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--envs", "128", "--rehearse", "1",
           "--dry-run", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = sorted((json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")), key=lambda r: r["rank"])
    assert [r["rank"] for r in rows] == [0, 1] and all(r["world"] == 2 for r in rows)
    assert [(r["env_lo"], r["env_hi"]) for r in rows] == [(0, 128), (128, 256)]
    ndev = max(1, torch.cuda.device_count())
    assert [r["device"] for r in rows] == [0 % ndev, 1 % ndev]


@pytest.mark.timeout(300)
def test_bench_eight_ranks_dry_run_shards_seeds_devices():
    """`bench.py --gpus 8 --dry-run 1`: the 8-GPU launch the driver will make, as far as it goes without GPUs -- eight
    ranks, contiguous 4096-env shards of 32768, seeds 1000 + global env index, device = local rank, 2 worlds per env."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "1"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = sorted((json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")), key=lambda r: r["rank"])
    assert [r["rank"] for r in rows] == list(range(8))
    for r in rows:
        assert (r["env_lo"], r["env_hi"]) == (4096 * r["rank"], 4096 * (r["rank"] + 1))
        assert (r["seed_lo"], r["seed_hi"]) == (1000 + 4096 * r["rank"], 1000 + 4096 * (r["rank"] + 1))
        assert r["n_worlds"] == 8192 and r["world"] == 8
    ndev = torch.cuda.device_count()
    assert [r["device"] for r in rows] == [i % max(1, ndev) for i in range(8)]


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples"))
    import ppo
    from gym_auv_amd import distributed as D
    D.init_from_env(backend="gloo")
    torch.manual_seed(0)                                   # same initial weights on both ranks (the example broadcasts rank 0's)
    net = ppo.ActorCritic(12)
    params = list(net.parameters())
    for prm in params:
        torch.distributed.broadcast(prm.data, 0)
    opt = torch.optim.Adam(params, lr=2e-4)
    torch.manual_seed(100 + rank)                          # ... different data
    o, a = torch.randn(64, 12), torch.randn(64, 2)
    loss = -net.log_prob(net.pi(o), a).mean() + net.v(o).pow(2).mean()
    opt.zero_grad()
    loss.backward()
    local = torch.cat([p.grad.reshape(-1) for p in params]).clone()
    ppo.average_gradients(params, world)
    avg = torch.cat([p.grad.reshape(-1) for p in params]).clone()
    torch.nn.utils.clip_grad_norm_(params, 0.5)
    opt.step()
    w = torch.cat([p.data.reshape(-1) for p in params])
    q.put((rank, local.tolist(), avg.tolist(), w.tolist()))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_ppo_data_parallel_gradients_agree_under_gloo():
    """The data-parallel branch of examples/ppo.py (gradient all-reduce) under gloo, world size 2: after one update the
    averaged gradients are the mean of the two ranks' local ones and the weights are identical on both ranks."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, l0, a0, w0), (_, l1, a1, w1) = res
    l0, l1, a0, a1 = map(torch.tensor, (l0, l1, a0, a1))
    assert not torch.allclose(l0, l1)                                        # the ranks really saw different data
    assert torch.allclose(a0, a1, rtol=0, atol=0) and torch.allclose(a0, 0.5 * (l0 + l1), rtol=1e-6, atol=1e-8)
    assert w0 == w1
