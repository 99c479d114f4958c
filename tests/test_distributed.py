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
    D.barrier()
    q.put((rank, lo, hi, {k: v.tolist() for k, v in full.items()}, t))
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
    for rank, lo, hi, full, t in res:
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
