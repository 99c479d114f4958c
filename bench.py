#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched gym-auv step() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher starts its N ranks itself (one process per GPU).

Workload (BASELINE.json configs[2], the configuration the >=1 M env-steps/s target is quoted
on): 4096 envs x 180 sensors per GPU, 50 static filled polygon obstacles per env, effective
reference dt = 0.5 s, LiDAR on, ColavRewarder, VecEnv auto-reset, i.i.d. U(action_space) actions
(torch seed 0) resident in HBM before the timed region.  The world bank holds --worlds-per-env
(default 2) worlds per environment: env g starts in the world of seed 1000 + g and an episode
that ends rebinds it to a world it has not seen (seed 1000 + k * total_envs + g for its k-th
episode, cyclically), as the reference regenerates the scenario on every reset
(envs/movingobstacles.py:28-95).  One "step" = one batched env.step() over all envs of the rank.
Weak scaling: per-GPU work is fixed as N grows.

The rank's envs are stepped as --sub-batches K sub-batches (default 4 at >= 2048 envs per rank): K independent launch
chains on K streams, no ordering between them inside the timed region (the actions are resident); they overlap on the
GPU -- one chain's sweeps run under another's dynamics chain and navigation tail.  Same envs, same results as K = 1,
bit for bit (tests/test_gpu_parity.py::test_sub_batches_bitwise).

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline      the step's dominant kernel: k_step_multi (T steps per launch: `avg_ms` = a run of back-to-back launches between one
                pair of HIP events on the chain's stream / their number, `alone_ms` = one launch by itself) or k_step_roles (one
                step per launch).  `kernels[...]`: its launch duration by HIP events on its own
                stream in this process (with sub-batches: while the other chains run beside it; what a kernel trace's
                average shows) and algorithmic bytes / that duration.  HBM leg (`achieved`, `frac`): with ONE chain exactly
                that figure; with K concurrent chains the K launches of a step run at the same time, so the leg is the
                algorithmic bytes of a whole step / the measured time per step of the timed region.  Algorithmic = what the
                implemented algorithm must touch, LiDAR segments counted for NEARBY obstacles only.  `traffic`: HBM bytes
                per step from the committed FETCH_SIZE / WRITE_SIZE passes; `valu`: VALU issue cycles per step from the
                committed SQ pass / (1024 SIMDs x clock x measured time per step), `wait_frac` = SQ_WAIT_ANY /
                SQ_WAVE_CYCLES; both carry the sha256 of the library they were measured on and `stale: true` when this
                process runs another one.  `bound`: the larger leg, or "latency" when neither reaches 0.6.
  cpu_baseline  the CPU oracle (a C port of the reference algorithm) timed on this box's host cores on a bounded sample of
                the same workload (median of 5 repeats of 500 steps), plus the reference's own
                Python step() as measured in the build container (a stated constant, never run here).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD = 256 * 4               # 256 CUs x 4 SIMD-32 (MI355X_MICROARCH.md)
CLOCK_PEAK_GHZ = 2.4           # max shader clock (MI355X_MICROARCH.md): the VALU leg's peak is N_SIMD x this


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--workload", default="polygons50", choices=["polygons50", "circles20", "moving28", "mixed47"])
    ap.add_argument("--worlds-per-env", type=int, default=2,
                    help="bank size / envs: 2 (default) lets every auto-reset land on a world the env has not seen; "
                         "1 = the round-1 bench (an env is reborn in its own world)")
    ap.add_argument("--fresh-worlds", type=int, default=0,
                    help="1: A FRESH WORLD ON EVERY RESET (auv_fresh_worlds_create; workload moving28 = the reference's MovingObstacles "
                         "scenario): --worlds-per-env bank slots per env, every slot an episode leaves is rebuilt on the device by a "
                         "refill pass beside the step path, so the generator's cost is INSIDE the timed region; `comparison` then "
                         "carries the rate of the same loop over a bank that just cycles")
    ap.add_argument("--fresh-period", type=int, default=16, help="--fresh-worlds: a refill pass every this many step calls")
    ap.add_argument("--fresh-batch", type=int, default=64, help="--fresh-worlds: worlds per refill pass at most")
    ap.add_argument("--multi", type=int, default=-1,
                    help="T: the open-loop loop enqueues T consecutive steps of every chain as ONE launch per chain (auv_step_multi: an "
                         "environment's step t + 1 starts when ITS step t is through, no barrier over the slice, no launch turn-around; "
                         "bit-identical to T single-step launches).  1: one launch per step and chain.  -1 (default): where the loop is open (api "
                         "pipelined, resident actions, no fresh worlds, no graph, --sub-batches 0) the shape is CALIBRATED before anything is "
                         "timed, in windows like the timed one: >= 256 steps -- one chain x 64-step launches against chains of one-step "
                         "launches; fewer -- one chain with the window as one / two launches or launches of 5 against the chains "
                         "(config.shape_calibration holds every candidate's rate); else 1")
    ap.add_argument("--multi-order", default="cohorts", choices=["cohorts", "steps"], help="--multi: workgroup order of a launch (include/auv_hip.h, auv_set_multi_order)")
    ap.add_argument("--multi-lead", type=int, default=16)
    ap.add_argument("--multi-lag", type=int, default=30)
    ap.add_argument("--graph", type=int, default=0,
                    help="0 (default): eager launches; K > 0: K consecutive steps captured in ONE hipGraph over the action "
                         "ring, one replay per K steps (steps not a multiple of K are finished eagerly)")
    ap.add_argument("--step-mode", default="auto", choices=["auto", "side_by_side", "one_launch"],
                    help="how a step is launched (include/auv_hip.h, AUV_STEP_*)")
    ap.add_argument("--actions", default="uniform", choices=["uniform", "pilot"],
                    help="uniform: i.i.d. U(action_space), resident in HBM (headline); pilot: closed loop, "
                         "a = (1, 0.15 * heading_error) computed on the device from the observation of the "
                         "previous step, so that episodes progress along the path (SURVEY 8(d), config 1)")
    ap.add_argument("--sub-batches", type=int, default=0,
                    help="K: the rank's envs are stepped as K contiguous sub-batches, each a launch chain on a stream of its "
                         "own (auv_step_pipelined); the chains overlap on the GPU -- same envs, same results, bit for bit.  "
                         "0 (default): 4 (the four compute pipes of the chip) when the rank has >= 2048 envs and the step "
                         "is launched eagerly from resident actions, else 1")
    ap.add_argument("--api", default="auto", choices=["auto", "pipelined", "async", "step"],
                    help="how the timed loop drives the env.  pipelined: open loop, auv_step_pipelined (the chains are not ordered "
                         "against each other or a consumer); async: VecEnv step_async + step_wait EVERY step (a full rendezvous of "
                         "all chains with the caller's stream, scripts/run.py:293-296); step: env.step() on the caller's stream "
                         "(one chain).  auto: pipelined for resident actions with sub-batches > 1, else step")
    ap.add_argument("--rendezvous", default="device", choices=["events", "device", "cp"],
                    help="--api async: how the chains are ordered against the caller's stream (include/auv_hip.h, AUV_RDV_*)")
    ap.add_argument("--inline-first", type=int, default=0,
                    help="--api async: 1 = the first sub-batch runs on the caller's stream, only the others on streams of their own")
    ap.add_argument("--one-graph", type=int, default=0,
                    help="--graph K with sub-batches > 1: 0 = one linear graph per chain on the chain's stream, 1 = ONE graph with a branch per chain")
    ap.add_argument("--probe-streams", type=int, default=1,
                    help="0: take any K streams for the sub-batch chains instead of K that were measured to run side by side "
                         "(for runs under a counter-collecting profiler, which serialises dispatches)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--procs", type=int, default=0, help="world-generation worker processes (0 = auto)")
    ap.add_argument("--bank-cache", default="auto",
                    help="path prefix: save / load the rank's generated world bank (.npz, keyed by workload, shard and bank size).  "
                         "'auto' (default): under the system temp directory, so that a repeated run (the driver benches N = 1, 2, 4, 8 "
                         "back to back) does not regenerate 8192 worlds per rank; '' = never cache")
    ap.add_argument("--rehearse", type=int, default=0,
                    help="1: allow more ranks than visible GPUs (ranks share devices, gloo instead of RCCL); the line "
                         "then reports n_gpus = distinct physical devices, not ranks")
    ap.add_argument("--dry-run", type=int, default=0,
                    help="1: every rank prints its shard (rank, env range, seeds, device) and exits without touching the GPU")
    return ap.parse_args()


WORKLOADS = {
    # name: (generator, kwargs, n_sectors, n_sensors_per_sector, description)
    "polygons50": ("polygon_world", dict(n_polygons=50), 9, 20,
                   "%d envs x 180 sensors per GPU, 50 static polygon obstacles (BASELINE configs[2])"),
    "circles20": ("static_circles_world", dict(n_circles=20), 8, 8,
                  "%d envs x 64 sensors per GPU, 20 static circular obstacles (BASELINE configs[1])"),
    "moving28": ("moving_obstacles_world", dict(), 9, 20,
                 "%d envs x 180 sensors per GPU, 17 moving + 11 static obstacles (BASELINE configs[3] shard)"),
    "mixed47": ("polygon_world", dict(n_polygons=10, n_circles=20, n_moving=17), 16, 16,
                "%d envs x 256 sensors per GPU, 20 circles + 10 polygons + 17 movers (BASELINE configs[4] shard)"),
}


def algorithmic_bytes(bank, S, world_of_env, nearby):
    """Per-launch algorithmic HBM bytes of each phase of the step: what the
    implemented algorithm has to touch, summed over the envs of one rank, fp64 layout.  The LiDAR
    sweep only reads the boundary segments of obstacles whose cached NEARBY flag is set
    (vessel.py:266-273), so only those are charged (`nearby` = the mask at the end of the timed
    run, [n, k_max])."""
    n = len(world_of_env)
    P = np.diff(bank["poly_off"])[world_of_env].astype(np.float64)
    obs_off = bank["obs_off"]
    K = np.diff(obs_off)[world_of_env].astype(np.float64)
    M = np.diff(bank["mv_off"])[world_of_env].astype(np.float64)
    meta = bank["obs_meta"]
    seg_near = np.zeros(n)
    k_max = nearby.shape[1]
    for j in range(k_max):                                 # obstacle slot j of every env
        has = j < np.diff(obs_off)[world_of_env]
        idx = np.minimum(obs_off[world_of_env] + j, len(meta) - 1)
        static = has & (meta[idx, 0] != 2) & (nearby[:, j] != 0)
        seg_near += np.where(static, meta[idx, 2], 0)
    S = float(S)
    k1 = 144.0 * n                                                          # state r/w, action, counter
    # reads: pose + counters + table descriptor, per obstacle meta / cull circle / nearby flag, the nearby
    # obstacles' segments (32 B each), per mover state r/w + parameters; writes: ranges, closeness (fp64 + f32),
    # cull limits, collision, reward term
    lidar = (104.0 + 41.0 * K + 32.0 * seg_near + 96.0 * M + 20.0 * S + 8.0 * K + 9.0).sum()
    nch = np.ceil((P - 1) / 64.0)
    nav = (32.0 * nch + 3 * 65 * 16.0 + 600.0).sum()                        # chunk circles + ~3 surviving chunks + knots/scalars
    reward = 300.0 * n                                                      # two reward terms, nav/info rows, counters, outputs
    all_seg = np.zeros(int(bank["n_worlds"]))
    w_of_obs = np.repeat(np.arange(int(bank["n_worlds"])), np.diff(obs_off))
    st = meta[:, 0] != 2
    np.add.at(all_seg, w_of_obs[st], meta[st, 2])
    return dict(k1=k1, lidar=float(lidar), nav=float(nav), reward=float(reward),
                nearby_segments_per_env=float(seg_near.mean()), all_segments_per_env=float(all_seg[world_of_env].mean()),
                nav_bruteforce=float((16.0 * P).sum()))


KERNEL_PHASES = {   # which phases of the step a launch performs (for its algorithmic byte count)
    "k1_dynamics": ("k1",), "k23_lidar_nav": ("lidar", "nav"), "k3_reward": ("reward",),
    "k_step_roles": ("k1", "lidar", "nav", "reward"),
}


def library_sha256():
    """sha256 of the HIP library this process runs (ties the committed counter passes to a binary)"""
    import hashlib
    from gym_auv_amd import _capi
    path = os.environ.get("AUV_HIP_LIB") or _capi.LIB_PATH
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (one per GPU,
    the shape of the reference's 8 SubprocVecEnv workers, scripts/run.py:293-296) with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, relay rank 0's JSON line, fail if any rank fails.  The parent never touches
    HIP (torch.cuda.device_count() does not initialise it on this image) and nothing is re-exec'd."""
    import socket
    import subprocess
    n = args.gpus
    ndev = torch.cuda.device_count()
    if ndev < n and not (args.rehearse or args.dry_run):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; one rank per GPU is required "
                         "(--rehearse 1 lets ranks share devices over gloo)" % (n, ndev))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    procs, files = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if args.rehearse and ndev < n:
            env.setdefault("AUV_DIST_BACKEND", "gloo")
        f = tempfile.TemporaryFile(mode="w+")
        files.append(f)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=f))
    # wait for all ranks; a rank that dies leaves the others in a collective, so end exactly those (by PID)
    failed = []
    while any(p.poll() is None for p in procs) and not failed:
        time.sleep(0.2)
        failed = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
    if failed:
        for q in procs:
            if q.poll() is None:
                q.kill()
    for p in procs:
        p.wait()
    failed = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
    outs = []
    for f in files:
        f.seek(0)
        outs.append(f.read())
        f.close()
    if failed:
        sys.stderr.write("bench.py: rank(s) failed: %s\n" % failed)
        return 1
    # relay the JSON line(s) only (gloo / RCCL may chat on a rank's stdout)
    lines = [l for o in (outs if args.dry_run else outs[:1]) for l in o.splitlines() if l.startswith("{")]
    sys.stdout.write("\n".join(lines) + "\n")
    sys.stdout.flush()
    return 0


def device_bank_summary(env):
    """What algorithmic_bytes needs of a bank that was built on the device (slot layout): vertex counts, obstacle records."""
    cnt = env.read_bank("POLY_CNT").cpu().numpy().astype(np.int64)
    meta = env.read_bank("OBS_META").cpu().numpy()
    W, K = meta.shape[0], meta.shape[1]
    M = env.m_max if env._gen.n_moving else 0
    return dict(n_worlds=W, poly_off=np.r_[0, np.cumsum(cnt)], obs_off=np.arange(W + 1, dtype=np.int64) * K,
                mv_off=np.arange(W + 1, dtype=np.int64) * M, obs_meta=meta.reshape(W * K, 4))


def world_seeds(lo, n_local, total_envs, worlds_per_env):
    """Seed of world w of this rank's bank: env g = lo + (w % n_local) meets it in its (w // n_local)-th episode."""
    w = np.arange(n_local * worlds_per_env)
    return 1000 + (w // n_local) * total_envs + lo + (w % n_local)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    ndev = torch.cuda.device_count()              # does not initialise HIP
    if ndev < world and not (args.rehearse or args.dry_run):
        raise SystemExit("WORLD_SIZE=%d but only %d GPU(s) visible (one rank per GPU; --rehearse 1 to share)" % (world, ndev))
    n_devices_used = len({r % max(1, ndev) for r in range(world)})   # ranks map to device LOCAL_RANK % ndev
    if args.rehearse and ndev < world:
        os.environ.setdefault("AUV_DIST_BACKEND", "gloo")            # (ranks sharing a device cannot use RCCL; also when an
                                                                     # outside launcher -- torch.distributed.run -- started them)

    # ---- reset-time host work first, BEFORE anything initialises the GPU (worker processes
    # are forked here; no process that has touched HIP forks or execs)
    from gym_auv_amd.config import effective_reference_config
    from gym_auv_amd.world import build_bank_parallel
    gen, kwargs, ns, nps, desc = WORKLOADS[args.workload]
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = ns, nps
    n_local = args.envs
    desc = desc % n_local
    from gym_auv_amd.distributed import shard_range
    lo, hi = shard_range(n_local * world, rank, world)     # weak scaling: fixed envs per GPU, contiguous blocks
    assert hi - lo == n_local
    wpe = max(1, args.worlds_per_env)
    seeds = world_seeds(lo, n_local, n_local * world, wpe)
    if args.dry_run:
        print(json.dumps(dict(rank=rank, world=world, env_lo=lo, env_hi=hi, seed_lo=int(seeds[0]), seed_hi=int(seeds[n_local - 1]) + 1,
                              n_worlds=len(seeds), device=int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev),
                              n_devices_used=n_devices_used)), flush=True)
        return
    procs = args.procs or max(1, min(16, host_cores() // max(1, world)))
    t0 = time.time()
    cache = ""
    fresh = bool(args.fresh_worlds)
    if fresh and args.workload != "moving28":
        raise SystemExit("--fresh-worlds builds the reference's MovingObstacles scenario on the device: --workload moving28")
    if fresh:
        wpe = 3 if args.worlds_per_env == 2 else max(2, wpe)     # (the mode's default depth; an explicit value is honoured)
    if args.bank_cache and not fresh:
        import hashlib
        import tempfile
        # the key names everything the bank depends on: workload, shard, bank size and the generator's own source
        src = hashlib.sha256()
        for mod in ("scenarios.py", "world.py", "obstacles.py", "path.py", "worldspec.py", "seeding.py"):
            src.update(open(os.path.join(ROOT, "gym_auv_amd", mod), "rb").read())
        prefix = os.path.join(tempfile.gettempdir(), "auv_bench_bank") if args.bank_cache == "auto" else args.bank_cache
        cache = "%s.%s.%d.%d.%d.%d.%s.npz" % (prefix, args.workload, lo, n_local, n_local * world, wpe, src.hexdigest()[:12])
    bank, bank_from_cache = None, False
    if cache and os.path.exists(cache):
        try:
            z = np.load(cache)
            bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
            bank_from_cache = True
        except Exception:
            bank = None                                    # (a torn file of an interrupted run: regenerate)
    if fresh:
        bank = None                                        # no host-side bank at all: the device builds every world
    elif bank is None:
        bank = build_bank_parallel(gen, seeds, procs=procs, **kwargs)
        if cache:
            tmp = "%s.%d.tmp.npz" % (cache, os.getpid())
            try:
                np.savez(tmp, **bank)
                os.replace(tmp, cache)                     # (atomic: another rank or run never reads half a file)
            except OSError:                                # (a full or read-only temp directory: the cache is a convenience)
                try:
                    os.remove(tmp)
                except OSError:
                    pass
    t_gen = time.time() - t0

    from gym_auv_amd import distributed as D
    rank, world, local = D.init_from_env()
    dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))   # (modulo: single-GPU rehearsal of N ranks)
    torch.cuda.set_device(dev)
    from gym_auv_amd.batched_env import BatchedAuvEnv     # fails loudly without the HIP library
    if fresh:
        from gym_auv_amd.devgen import FreshWorlds
        t0 = time.time()
        # the world of env g's k-th episode = f(seed, GLOBAL env index, k): the shard's first global index is its base
        env = BatchedAuvEnv(cfg, FreshWorlds(depth=wpe, seed=1000, env_index_base=lo, batch_cap=args.fresh_batch, period=args.fresh_period),
                            n_local, device=dev, auto_reset=True)
        torch.cuda.synchronize(dev)
        t_gen = time.time() - t0
        bank = device_bank_summary(env)                    # counts / offsets / obstacle records of the slots, for the byte model
    else:
        env = BatchedAuvEnv(cfg, bank, n_local, device=dev, auto_reset=True)
    env.set_step_mode(args.step_mode)
    S = env.n_sensors

    g = torch.Generator(device=dev)
    g.manual_seed(0 + rank)
    n_pool = 64
    low = torch.tensor([-1.0, -0.15], device=dev)
    high = torch.tensor([1.0, 0.15], device=dev)
    pool = low + (high - low) * torch.rand((n_pool, n_local, 2), generator=g, device=dev)   # resident in HBM

    env.reset()
    K = max(0, args.graph)
    sub = args.sub_batches
    api = args.api
    if api == "auto":
        api = "step" if (args.actions == "pilot" and sub <= 1) else "pipelined"
    # several steps per launch (auv_step_multi) where the loop is open and long enough to fill the pipeline of a launch: ONE chain of
    # 64-step launches (measured, tools/multi_sweep.sh: 170 M env-steps/s against 153 M for four chains of one-step launches over
    # 2000 steps; the driver's 20-step window as ONE launch: 157-164 M against 132-138 M for four one-step chains -- once the
    # multi-step kernel is warm: its first launch inside the timed region had cost 0.15 ms and hidden this)
    open_loop = api == "pipelined" and args.actions == "uniform" and not fresh and not K
    multi_T = args.multi
    calibration = None
    if multi_T < 0:
        multi_T = 1
        if open_loop and args.steps >= 4 and n_local % 64 == 0 and env.effective_step_mode(n_local) == "one_launch" and args.sub_batches == 0:
            # which open-loop shape is faster for THIS workload and THIS window?  Measured before anything is timed, in windows shaped
            # like the timed region (synchronize on both sides).  Long runs (>= 256 steps): 192 steps of one chain x 64-step launches
            # (170 against 153 M env-steps/s at 4096 x 180) and of four chains x one-step launches (141 against 125 M at 8192 x 256:
            # its sweeps' larger LDS slices leave a launch of several steps too few wave slots to pipeline in).  Short runs (the
            # driver's 20 steps): the whole window as ONE launch, as two, in launches of five, or as four one-step chains -- a
            # launch's ramp and drain weigh as much as its steady state there.
            def trial(k, T, n, reps):
                env.set_sub_batches(k, probe_streams=bool(args.probe_streams))
                env.set_multi_order(args.multi_order, args.multi_lead, args.multi_lag)
                dts = []
                for rep in range(reps + 1):                          # (the first window warms the shape's kernel up)
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    for i in range(0, n, T):
                        if T > 1:
                            env.step_multi(pool, i % n_pool, T)
                        else:
                            env.step_pipelined(pool[i % n_pool])
                    torch.cuda.synchronize(dev)
                    dts.append(time.perf_counter() - t1)
                return n_local * n / float(np.median(dts[1:]))
            want4 = n_local >= 2048 and env.effective_step_mode(n_local // 4) == "one_launch"
            if args.steps >= 256:
                r_multi, r_chains = trial(1, 64, 192, 1), trial(4 if want4 else 1, 1, 192, 1)
                multi_T = 64 if r_multi > r_chains else 1
                calibration = dict(one_chain_64_steps_per_launch=round(r_multi, 1), chains_one_step_per_launch=round(r_chains, 1), steps_each=192)
            else:
                divs = [T for T in range(min(64, n_pool, args.steps), 1, -1) if args.steps % T == 0]      # whole launches only
                cands = sorted(set(divs[:2] + [T for T in (5,) if T in divs]), reverse=True)
                rates = {T: trial(1, T, args.steps, 5) for T in cands}
                rates[1] = trial(4 if want4 else 1, 1, args.steps, 5)
                multi_T = max(rates, key=rates.get)
                calibration = dict({("one_chain_%d_steps_per_launch" % T if T > 1 else "chains_one_step_per_launch"): round(r, 1) for T, r in rates.items()},
                                   steps_each=args.steps, windows_each=5)
            env.reset()
    if multi_T > 1 and not open_loop:
        raise SystemExit("--multi T: open-loop stretches only (api pipelined, resident actions, no --fresh-worlds, no --graph)")
    if sub <= 0:
        want4 = n_local >= 2048 and env.effective_step_mode(n_local // 4) == "one_launch"
        sub = 4 if (want4 and api in ("pipelined", "async") and args.actions == "uniform") else 1
        if sub == 4 and fresh:
            sub = 3      # (at most four kernels run side by side on this GPU: three chains and the refill passes' stream)
        if multi_T > 1:
            sub = 1      # (a launch of several steps pipelines the whole batch by itself)
    if api == "step":
        sub = 1
    t_probe = 0.0
    if sub > 1:
        # fewer streams side by side than chains asked for would silently time a different shape: fail instead
        for attempt in range(3):
            try:
                env.set_sub_batches(sub, probe_streams=bool(args.probe_streams), inline_first=bool(args.inline_first and api == "async"),
                                    strict=bool(args.probe_streams))
                break
            except RuntimeError:
                # (the stream probe times kernels against the wall clock; with several ranks starting on one host it can read a
                # busy host as "serialised": ask again before giving up -- and do give up rather than time another shape)
                if attempt == 2:
                    raise
                time.sleep(0.5 * (attempt + 1))
        t_probe = env.stream_probe_s
    elif api == "async":
        env.set_sub_batches(1, inline_first=bool(args.inline_first))
    if args.probe_streams or sub <= 1:
        env.rendezvous = args.rendezvous        # (unprobed streams -- profiler runs -- keep the event-based ordering set_sub_batches chose)
    act = torch.ones((n_local, 2), dtype=torch.float32, device=dev)

    def pilot_all():
        # look-ahead pilot: full thrust, rudder proportional to the heading error (observation column 4, already clipped
        # to +-1 rad): a tiny torch kernel per step on the caller's stream, all on the device
        torch.mul(env.obs[:, 4], 0.15, out=act[:, 1])

    if K:
        # the 64 pre-generated action batches become the action ring of the captured steps: a replay consumes
        # the next K slots, nothing is copied or re-bound in the timed loop.  Steps beyond a multiple of K are
        # launched eagerly from the pool (eager launches never touch the ring).
        if n_pool % K and K % n_pool:
            raise SystemExit("--graph K: K must divide %d (the action pool) or be a multiple of it" % n_pool)
        if args.actions != "uniform":
            raise SystemExit("--graph replays open-loop stretches: --actions uniform")
        if sub > 1:
            # captured CHAINS: K steps of every sub-batch per replay, each chain with its own position in the ring
            ring = env.capture_graph_chains(torch.float32, slots=n_pool, steps=K, one_graph=bool(args.one_graph))
            if args.steps % K or args.warmup % K:
                # every chain keeps its own ring position and eager steps in between would not advance it: whole replays only.
                # Rounded UP (the default warm-up of 200 made `--graph 16` exit); the JSON line reports what ran.
                args.steps, args.warmup = -(-args.steps // K) * K, -(-args.warmup // K) * K
                sys.stderr.write("bench.py: --graph %d with sub-batch chains: steps / warmup rounded up to %d / %d (whole replays)\n"
                                 % (K, args.steps, args.warmup))
        else:
            ring = env.capture_graph(torch.float32, slots=n_pool, steps=K)
        ring.copy_(pool)
        api = "graph"

        def run(i0, n):
            for _ in range(n // K):
                env.step_graph()
            for i in range(n % K):
                env.step(pool[(i0 + i) % n_pool])
    elif api == "async":
        # VecEnv protocol, a FULL rendezvous per step: every chain waits for the caller's stream, the caller's stream for
        # every chain (scripts/run.py:293-296: SubprocVecEnv.step_async / step_wait)
        def run(i0, n):
            for i in range(n):
                if args.actions == "pilot":
                    pilot_all()
                    env.step_async(act)
                else:
                    env.step_async(pool[(i0 + i) % n_pool])
                env.step_wait()
    elif args.actions == "pilot" and sub > 1:
        def run(i0, n):
            # the pilot of a sub-batch runs on that sub-batch's stream, so the chains stay independent of each other: a
            # caller restructured around the chains (examples/ppo.py), NOT the VecEnv protocol
            for _ in range(n):
                for i, (lo, cnt) in enumerate(env._slices):
                    with torch.cuda.stream(env._sub_streams[i]):
                        torch.mul(env.obs[lo:lo + cnt, 4], 0.15, out=act[lo:lo + cnt, 1])
                    env.step_slice(i, act)
        api = "per_chain_pilot"
    elif args.actions == "pilot":
        def run(i0, n):
            for _ in range(n):
                pilot_all()
                env.step(act)
    elif multi_T > 1:
        # several steps per launch and chain: the action pool is the ring, step i reads slot i % n_pool
        if env._slices is None or env.sub_batches != sub:
            env.set_sub_batches(sub, probe_streams=bool(args.probe_streams))
        env.set_multi_order(args.multi_order, args.multi_lead, args.multi_lag)
        api = "multi"

        def run(i0, n):
            i = 0
            while n - i >= multi_T:
                env.step_multi(pool, (i0 + i) % n_pool, multi_T)
                i += multi_T
            for j in range(i, n):                                  # (a remainder shorter than T: single-step launches)
                env.step_pipelined(pool[(i0 + j) % n_pool])
    elif sub > 1:
        def run(i0, n):
            # K independent launch chains: sub-batch s of step i goes to stream s; nothing orders the chains against
            # each other (the actions are resident), the synchronize() around the timed region waits for all of them
            # (delaying chain i once by i / K of a step period when a stretch starts -- chains that start together might
            # run in lockstep -- was measured: no difference at 20, 100, 500 steps; they are out of phase by themselves)
            for i in range(n):
                env.step_pipelined(pool[(i0 + i) % n_pool])
    else:
        def run(i0, n):
            for i in range(n):
                env.step(pool[(i0 + i) % n_pool])
    sub = env.sub_batches if env._slices is not None else 1

    run(0, args.warmup)
    torch.cuda.synchronize(dev)
    D.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    torch.cuda.synchronize(dev)
    # every rank reads its clock when ITS steps are complete, then all meet at the closing barrier and the slowest rank's
    # time counts (max over ranks).  Reading the clock behind the barrier instead would add one RCCL barrier (tens of
    # microseconds over xGMI) to every rank's time -- a tenth of the driver's 0.7 ms window of 20 steps, and nothing
    # the step path does: the path has no collective.
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    D.barrier()
    torch.cuda.synchronize(dev)
    elapsed = D.max_over_ranks(elapsed, dev)
    # the only exchange of the path: episode returns for reporting, RCCL all_gather over xGMI -- timed on its own (second call:
    # the first one pays the communicator's lazy set-up), outside the timed region
    ep_stats = env.episode_stats()
    stats = D.gather_episode_stats(ep_stats)
    torch.cuda.synchronize(dev)
    t_ag = time.perf_counter()
    stats = D.gather_episode_stats(ep_stats)
    torch.cuda.synchronize(dev)
    t_ag = time.perf_counter() - t_ag
    total_envs = n_local * world
    value = total_envs * args.steps / elapsed
    if api == "pipelined" and sub == 1:
        api = "step"
    per_rank = D.gather_floats([t_gen, t_probe, float(bank_from_cache), float(sub), 1e3 * elapsed_local / args.steps, 1e3 * t_ag], dev)

    world_of_env = env.read("WORLD_IDX").cpu().numpy()
    nearby = env.read("NEARBY").cpu().numpy()

    # ---- per-kernel timing with HIP events on the launch stream(s) (same workload, same process)
    n_prof = min(max(args.steps, 20), 100)
    alg = algorithmic_bytes(bank, S, world_of_env, nearby)
    step_bytes = sum(alg[ph] for ph in ("k1", "lidar", "nav", "reward"))
    pipelined = sub > 1 and env.effective_step_mode(max(1, n_local // sub)) == "one_launch"
    if api == "multi":
        # a run of back-to-back launches of multi_T steps bracketed by ONE pair of HIP events per chain, on that chain's stream (the
        # stream is in order: the pair encloses exactly those launches, as the timed region issues them) => average launch duration;
        # `alone_ms`: one launch between its own pair of events (its ramp and drain not hidden under its neighbours)
        n_launch = max(4, min(max(args.steps, 1024) // multi_T, 24))      # (16 launches of 64 steps; the driver's 20 steps: 24 launches of 20)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in env._sub_streams]
        ev1 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in env._sub_streams]
        for (a, _), stq in zip(evs, env._sub_streams):
            a.record(stq)
        for j in range(n_launch):
            env.step_multi(pool, (j * multi_T) % n_pool, multi_T)
        for (_, b), stq in zip(evs, env._sub_streams):
            b.record(stq)
        torch.cuda.synchronize(dev)
        for (a, _), stq in zip(ev1, env._sub_streams):
            a.record(stq)
        env.step_multi(pool, 0, multi_T)
        for (_, b), stq in zip(ev1, env._sub_streams):
            b.record(stq)
        torch.cuda.synchronize(dev)
        lms = np.array([a.elapsed_time(b) / n_launch for a, b in evs])      # ms per launch, per chain
        alone = np.array([a.elapsed_time(b) for a, b in ev1])
        names = ["k_step_multi"]
        kms = np.array([lms.mean(), 0.0, 0.0, lms.max()])
        launch_bytes = step_bytes / env.sub_batches * multi_T
        per_kernel = {"k_step_multi": dict(avg_ms=round(float(lms.mean()), 5), steps_per_launch=multi_T, launches_per_step=round(env.sub_batches / multi_T, 4),
                                           algorithmic_bytes=int(launch_bytes),
                                           achieved_GBs=round(launch_bytes / (lms.mean() * 1e-3) / 1e9, 1),
                                           frac=round(launch_bytes / (lms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           per_slice_ms=[round(float(x), 5) for x in lms], launches_timed=n_launch,
                                           alone_ms=round(float(alone.mean()), 5),
                                           avg_ms_per_step=round(float(lms.mean()) / multi_T, 5))}
    elif pipelined:
        # every sub-batch launch stamped on its own stream while the other chains run beside it
        lms = np.zeros(env.sub_batches)
        for i in range(n_prof):
            lms += np.array(env.step_pipelined_timed(pool[i % n_pool]))
        lms /= n_prof
        names = ["k_step_roles"]
        kms = np.array([lms.mean(), 0.0, 0.0, lms.max()])
        launch_bytes = step_bytes / env.sub_batches
        per_kernel = {"k_step_roles": dict(avg_ms=round(float(lms.mean()), 5), launches_per_step=env.sub_batches,
                                           algorithmic_bytes=int(launch_bytes),
                                           achieved_GBs=round(launch_bytes / (lms.mean() * 1e-3) / 1e9, 1),
                                           frac=round(launch_bytes / (lms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           per_slice_ms=[round(float(x), 5) for x in lms])}
    else:
        kms = np.zeros(4)
        for i in range(n_prof):
            kms += np.array(env.step_timed(pool[i % n_pool]))
        kms /= n_prof
        names = env.timed_kernel_names()
        per_kernel = {}
        for j, nm in enumerate(names):
            b = sum(alg[ph] for ph in KERNEL_PHASES[nm])
            gbs = b / (kms[j] * 1e-3) / 1e9 if kms[j] > 0 else 0.0
            per_kernel[nm] = dict(avg_ms=round(float(kms[j]), 5), launches_per_step=1, algorithmic_bytes=int(b),
                                  achieved_GBs=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
    dom = max(names, key=lambda nm: per_kernel[nm]["avg_ms"])
    per_kernel[dom].update(lidar_bytes=int(alg["lidar"]), nav_bytes=int(alg["nav"]),
                           nearby_segments_per_env=round(alg["nearby_segments_per_env"], 1),
                           all_segments_per_env=round(alg["all_segments_per_env"], 1),
                           nav_bruteforce_bytes=int(alg["nav_bruteforce"]))
    # HBM leg.  One chain: algorithmic bytes of the dominant launch / its HIP-event duration.  Sub-batch chains: the
    # launches of a step run CONCURRENTLY, so one launch's bytes / duration is a quarter of what the chip moves; the
    # leg is then all launches' bytes of a step / the measured time per step of the timed region (gaps included).
    ms_step = 1e3 * elapsed / args.steps
    if sub > 1:
        achieved = step_bytes / (ms_step * 1e-3) / 1e9
    else:
        achieved = per_kernel[dom]["achieved_GBs"]
    hbm_frac = round(achieved / HBM_PEAK_GBS, 4)

    # ---- the same envs through the OTHER ways of driving them, beside the headline (rank 0 at N = 1; short, outside
    # the timed region): the headline loop is open-loop chain throughput, a VecEnv consumer that waits for every step
    # sees the rendezvous figure
    comparison = None
    if rank == 0 and world == 1 and args.actions == "uniform" and not K and args.steps >= 20 and args.probe_streams:
        n_cmp = 300

        def rate(fn):
            fn(40)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            fn(n_cmp)
            torch.cuda.synchronize(dev)
            return round(n_local * n_cmp / (time.perf_counter() - t1), 1)

        def loop_step(n):
            for i in range(n):
                env.step(pool[i % n_pool])

        def loop_async(n):
            for i in range(n):
                env.step_async(pool[i % n_pool])
                env.step_wait()

        comparison = dict(steps=n_cmp)
        if not fresh:
            # (with a fresh world per reset the refill passes' stream was chosen to share no hardware queue with the CHAINS' streams;
            # the caller's own stream, which these two loops run on, may well share one with it: not a figure of that mode)
            comparison["one_chain_step"] = rate(loop_step)
            if env._slices is None:
                env.set_sub_batches(1, inline_first=True)
            comparison["step_async_wait_%s_sub%d" % (env.rendezvous, env.sub_batches)] = rate(loop_async)
        # the chains of one-step launches and the policy-in-the-loop figure below on FOUR chains also when the timed loop ran one
        # chain of several steps per launch (the driver's record then carries both)
        k_cmp = sub
        if sub == 1 and not fresh and n_local >= 2048 and env.effective_step_mode(n_local // 4) == "one_launch":
            k_cmp = 4
        if k_cmp > 1:
            def loop_pipe(n):
                for i in range(n):
                    env.step_pipelined(pool[i % n_pool])
            try:
                if env._slices is None or env.sub_batches != k_cmp:
                    env.set_sub_batches(k_cmp, probe_streams=True)
                comparison["pipelined_sub%d" % env.sub_batches] = rate(loop_pipe)
            except Exception as exc:
                comparison["pipelined_error"] = repr(exc)[:200]
        if not fresh and n_local % 64 == 0 and env.effective_step_mode(n_local) == "one_launch":
            # the sustained open-loop rate with 64 steps per launch on one chain, whatever the timed loop above was (the driver's
            # 20-step window cannot show it): 640 steps after 128 of warm-up, outside the timed region
            try:
                env.set_sub_batches(1)
                env.set_multi_order(args.multi_order, args.multi_lead, args.multi_lag)
                for j in range(2):
                    env.step_multi(pool, (64 * j) % n_pool, 64)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for j in range(10):
                    env.step_multi(pool, (64 * j) % n_pool, 64)
                torch.cuda.synchronize(dev)
                comparison["open_loop_multi_T64_sub1"] = round(n_local * 640 / (time.perf_counter() - t1), 1)
                if k_cmp > 1:
                    env.set_sub_batches(k_cmp, probe_streams=True)
            except Exception as exc:
                comparison["open_loop_multi_error"] = repr(exc)[:200]
        # the closed loop a PPO run lives in (scripts/run.py:332-357: MlpPolicy [256, 128, 64] for policy and value): the fused
        # policy launch (csrc/k6_policy.hip, exact f32 on the matrix cores, random-init weights) and the environment's step of
        # every chain back to back, T transitions stored per environment, one C call (auv_policy_rollout) -- so that a
        # driver-run record carries the figure a learner sees
        try:
            sys.path.insert(0, os.path.join(ROOT, "examples"))
            import ppo as ppo_example
            from gym_auv_amd.policy import FusedActorCritic
            if env._slices is None:
                env.set_sub_batches(1)
            torch.manual_seed(0)
            net = ppo_example.ActorCritic(env.obs_dim).to(dev)
            T_roll = 128
            fused = FusedActorCritic(net, env, rollout=T_roll, reward_scale=0.01)
            fused.begin_rollout()
            fused.rollout(16, flush=False)
            torch.cuda.synchronize(dev)
            fused.begin_rollout()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            fused.rollout(T_roll)
            torch.cuda.synchronize(dev)
            comparison["policy_rollout_sub%d" % env.sub_batches] = round(n_local * T_roll / (time.perf_counter() - t1), 1)
            comparison["policy_rollout_steps"] = T_roll
        except Exception as exc:                              # (reported, never fatal: the headline does not depend on it)
            comparison["policy_rollout_error"] = repr(exc)[:200]

    fresh_stats = None
    if fresh:
        torch.cuda.synchronize(dev)
        fresh_stats = env.fresh_stats()
        if rank == 0 and world == 1 and not K and args.probe_streams:
            # beside it: the SAME loop (same steps, same chains) over a device-built bank of the same shape that just cycles
            # (w + N) % W -- the difference is what a fresh world per reset costs, generator and refill bookkeeping included
            from gym_auv_amd.devgen import GeneratedWorlds
            cyc = BatchedAuvEnv(cfg, GeneratedWorlds(wpe * n_local, seed=1000), n_local, device=dev, auto_reset=True)
            cyc.set_step_mode(args.step_mode)
            cyc.reset()
            if sub > 1:
                cyc.set_sub_batches(sub, probe_streams=True, strict=True)

            def loop_cyc(n):
                for i in range(n):
                    if sub > 1:
                        cyc.step_pipelined(pool[i % n_pool])
                    else:
                        cyc.step(pool[i % n_pool])
            loop_cyc(args.warmup)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            loop_cyc(args.steps)
            torch.cuda.synchronize(dev)
            t_cyc = time.perf_counter() - t1
            comparison = dict(comparison or {}, bank_cycling_same_loop=round(n_local * args.steps / t_cyc, 1),
                              fresh_worlds_timed_region=round(value, 1), steps_each=args.steps,
                              bank_cycling_episodes=int(cyc.episode_stats()["episodes"].sum().item()))
            cyc.close()

    lib_sha = library_sha256()
    # the committed counter passes: of this very shape ("polygons50/sub1_T64": MULTI=64 tools/pmc_workload.sh) when there is one,
    # otherwise of one step per launch over as many chains (k_step_roles: the device functions k_step_multi runs)
    cfg_key = "%s/sub%d" % (args.workload, sub)
    cfg_key_T = cfg_key + ("_T%d" % multi_T if multi_T > 1 else "")

    def committed(name):
        """per-STEP counters of the step's launches from the committed rocprofv3 passes (4096 envs per GPU, this
        workload and sub-batch count), with the sha256 of the library they were measured on"""
        path = os.path.join(ROOT, "profiles", name)
        if n_local != 4096 or not os.path.exists(path):
            return None
        try:
            tab = json.load(open(path))
            k = cfg_key_T if cfg_key_T in tab else cfg_key
            if multi_T > 1 and cfg_key_T not in tab:
                # (no pass of this very launch length: the counters per STEP of another length of the same kernel, nearest first)
                others = sorted((abs(int(x.rsplit("_T", 1)[1]) - multi_T), x) for x in tab if x.startswith(cfg_key + "_T"))
                if others:
                    k = others[0][1]
            return dict(tab[k], key=k) if k in tab else None
        except Exception:
            return None

    traffic = committed("pmc_traffic.json")
    sq = committed("pmc_sq.json")
    valu = None
    if traffic is not None:
        traffic = dict(traffic, stale=traffic.get("lib_sha256") != lib_sha)
    if sq:
        # SQ_ACTIVE_INST_VALU counts quad-cycles (4 shader cycles) in which a wave executes a VALU instruction, summed over
        # waves and (here) over the launches of one step.  Issue capacity of a step = 1024 SIMDs x shader clock x time per
        # step; the clock is the counter pass's own SQ_BUSY_CYCLES / 32 shader engines / kernel duration.
        # `frac` prices the step against the PEAK clock (a lower bound of the share: the clock under load is lower);
        # `frac_at_measured_clock` against the counter pass's own SQ_BUSY_CYCLES / 32 shader engines / dispatch duration
        # (an upper bound: short launches do not keep every shader engine busy from first to last cycle).
        issue = 4.0 * sq["SQ_ACTIVE_INST_VALU"]
        cyc_step = CLOCK_PEAK_GHZ * 1e9 * ms_step * 1e-3
        valu = dict(insts=int(sq["SQ_INSTS_VALU"]), issue_cycles=int(issue), step_cycles=int(cyc_step), clock_ghz=CLOCK_PEAK_GHZ,
                    frac=round(issue / (N_SIMD * cyc_step), 4),
                    frac_at_measured_clock=round(issue / (N_SIMD * sq["clock_ghz"] * 1e9 * ms_step * 1e-3), 4), measured_clock_ghz=sq["clock_ghz"],
                    wait_frac=round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 4) if sq.get("SQ_WAVE_CYCLES") else None,
                    source="profiles/pmc_sq.json [%s]" % sq["key"], lib_sha256=sq.get("lib_sha256"),
                    stale=sq.get("lib_sha256") != lib_sha)
    legs = dict(hbm=hbm_frac, valu=valu["frac"] if valu else 0.0)
    bound = max(legs, key=legs.get)
    if max(legs.values()) < 0.6:
        bound = "latency"      # neither leg saturated: the step is bound by dependent chains / occupancy, not by a pipe
    # (achieved / peak / unit / frac are ALWAYS the HBM leg -- algorithmic bytes over measured time against 8 TB/s; `bound` names the
    # larger of `legs`, or "latency" when neither reaches 0.6)
    roofline = dict(bound=bound, kernel=dom, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=hbm_frac, frac_is_leg="hbm",
                    traffic=traffic, valu=valu, wait_frac=valu["wait_frac"] if valu else None, legs=legs,
                    step_bytes=int(step_bytes), concurrent_launches=sub if (pipelined or api == "multi") else 1, kernels=per_kernel,
                    lib_sha256=lib_sha)

    loops = {"pipelined": "open loop: %d chains, no ordering between them or with a consumer (auv_step_pipelined)" % sub,
             "multi": "open loop: %d chains, %d consecutive steps per launch and chain (auv_step_multi; bit-identical to single-step launches)" % (sub, multi_T),
             "step": "env.step() on the caller's stream, one launch per step",
             "async": "VecEnv step_async + step_wait every step: a full rendezvous of %d chain(s) with the caller's stream (%s%s)"
                      % (sub, args.rendezvous, ", first chain on the caller's stream" if args.inline_first else ""),
             "graph": "open loop: %d captured step(s) per replay, %d chain(s)%s" % (K, sub, ", one graph" if (args.one_graph and sub > 1) else ""),
             "per_chain_pilot": "closed loop PER CHAIN: %d chains, each with its pilot on its own stream (no rendezvous)" % sub}
    shape = env.effective_step_mode(max(1, n_local // sub))
    if K > 1 and sub == 1:
        # a captured graph of several steps over the whole batch replays the three-launch shape with its fused
        # reward + dynamics launch whatever the handle's mode is (include/auv_hip.h)
        shape = "side_by_side+k31 (captured, %d steps per graph)" % K
    cfg_out = dict(workload=desc, envs_per_gpu=n_local, n_sensors=S, total_envs=total_envs,
                   parallelism="env-shard x%d (no step-path collective)" % world, ranks=world,
                   hipgraph_steps=K, steps_per_launch=multi_T, shape_calibration=calibration, sub_batches=sub, api=api, loop=loops[api],
                   step_mode=shape, roofline_step_mode=env.effective_step_mode(max(1, n_local // sub)), actions=args.actions,
                   worlds_per_env=wpe, world_gen_s=round(t_gen, 1),
                   # first contact with an 8-GPU node: what every rank spent before the timed region, and on what
                   per_rank=dict(world_gen_s=[round(r[0], 1) for r in per_rank], stream_probe_s=[round(r[1], 2) for r in per_rank],
                                 bank_from_cache=[int(r[2]) for r in per_rank], sub_batches=[int(r[3]) for r in per_rank],
                                 # every rank's OWN time per step (the line's ms_per_step is their maximum) and what the one
                                 # reporting collective took on it: a first SCALE record that explains itself
                                 ms_per_step=[round(r[4], 5) for r in per_rank], episode_all_gather_ms=[round(r[5], 3) for r in per_rank]),
                   collective_backend=D.backend_name(),
                   episodes_finished=int(stats["episodes"].sum().item()))
    if fresh_stats:
        # `reused` must be 0: every finished episode found an unseen world waiting; `regenerated`: worlds rebuilt by the refill
        # passes (warm-up included), i.e. the generator ran inside the timed region
        cfg_out["fresh_worlds"] = dict(fresh_stats, period=args.fresh_period, worlds="(seed 1000, global env index, episode serial)")
    if os.environ.get("AUV_HIP_LIB"):
        cfg_out["lib_override"] = os.environ["AUV_HIP_LIB"]      # (the loader's A/B hook: say so when it is in use)
    out = dict(metric="env-steps/sec", value=round(value, 1), unit="env-steps/s", n_gpus=n_devices_used, steps=args.steps,
               warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 5), higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f64", data="synthetic", config=cfg_out, roofline=roofline)
    if comparison:
        out["comparison"] = comparison

    if rank == 0 and world == 1 and args.cpu_baseline:
        if fresh:
            # (the oracle steps host-built worlds of the same scenario: the reference's generator through scenarios.py)
            cpu_bank = build_bank_parallel(gen, world_seeds(0, 1024, 1024, 2), procs=procs, **kwargs)
            out["cpu_baseline"] = cpu_baseline(cfg, cpu_bank, min(1024, n_local))
        else:
            out["cpu_baseline"] = cpu_baseline(cfg, bank, n_local)
    if rank == 0:
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        torch.distributed.destroy_process_group()


def cpu_baseline(cfg, bank, n_local):
    """The CPU oracle (C port of the reference algorithm, fp64, -O2) on this box's host cores, on a bounded sample of
    the same workload, measured as BASELINE.md section 3 asks: warm-up, then the median of 5 repeats of >= 500 batched
    steps -- the first 1024 envs/worlds on all usable cores (OpenMP over envs), the first 256 on 1 thread (~20 s in
    all).  Beside it, as a stated constant, the reference's OWN Python step() timed in the build container
    (oracle/ref_harness/time_reference.py -> reference_timing.json; it cannot run on the GPU box)."""
    from gym_auv_amd._capi import make_config
    from oracle import pyoracle
    rs = np.random.RandomState(0)
    steps, repeats = 500, 5

    def leg(n, threads):
        ora = pyoracle.Oracle(make_config(cfg, auto_reset=True), n, bank)
        acts = rs.uniform([-1, -0.15], [1, 0.15], (8, n, 2))
        used = pyoracle.set_threads(threads)
        ora.reset()
        for i in range(50):
            ora.step(acts[i % 8])
        rates = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            for i in range(steps):
                ora.step(acts[i % 8])
            rates.append(n * steps / (time.perf_counter() - t0))
        return used, float(np.median(rates)), [round(r, 1) for r in rates]

    n_all, n_one = min(1024, n_local), min(256, n_local)
    cores, v_all, r_all = leg(n_all, host_cores())
    _, v_one, r_one = leg(n_one, 1)
    pyoracle.set_threads(cores)
    res = dict(value=round(v_all, 1), unit="env-steps/s", cores=cores, kind="port",
               sample="median of %d repeats of %d steps after 50 warm-up steps: %d envs of the same workload on %d threads "
                      "(OpenMP over envs); 1 thread: %d envs" % (repeats, steps, n_all, cores, n_one),
               repeats=r_all, value_1thread=round(v_one, 1), repeats_1thread=r_one)
    ref = os.path.join(ROOT, "oracle", "ref_harness", "reference_timing.json")
    if os.path.exists(ref):
        try:
            res["reference_python"] = json.load(open(ref))
        except Exception:
            pass
    return res


if __name__ == "__main__":
    main()
