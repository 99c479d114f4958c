"""BatchedAuvEnv — N independent gym-auv environments advanced per call by one HIP launch (or one per sub-batch).

VecEnv-shaped surface (what the reference's callers use through stable-baselines' VecEnv,
/root/reference/scripts/run.py:293-296): `reset() -> obs[N, D]`,
`step(actions[N, 2]) -> (obs[N, D], reward[N], done[N], info)`, auto-reset of finished
episodes.  All tensors are torch tensors on the env's GPU; nothing is copied to the host
inside `step`.  The per-environment semantics are those of BaseEnvironment.reset/step
(/root/reference/gym_auv/environment.py:176-366).

The HIP extension is mandatory: importing this module dlopens gym_auv_amd/csrc/libauv_hip.so
and raises `AuvLibraryError` if it is missing.  There is no CPU or eager-PyTorch fallback.
"""
import ctypes as C
import time
from typing import Dict, Optional, Sequence, Union

import numpy as np
import torch

from . import _capi
from ._capi import FIELD_DTYPES, FIELDS, AuvLibraryError, load_library, make_bank_struct, make_config
from .config import Config
from .devgen import FreshWorlds, GeneratedWorlds
from .spaces import Box
from .world import BuiltWorld, build_world, pack_bank
from .worldspec import WorldSpec

_LIB = load_library()   # fail loudly at import time

_TORCH_DTYPES = {np.float64: torch.float64, np.int32: torch.int32, np.uint8: torch.uint8, np.int64: torch.int64}


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, _LIB.auv_last_error().decode()))


class BatchedAuvEnv:
    def __init__(self, config: Config, worlds: Union[Dict[str, np.ndarray], Sequence[Union[WorldSpec, BuiltWorld]]],
                 n_envs: int, device: Union[str, torch.device] = "cuda:0", rewarder: str = "colav",
                 test_mode: bool = False, cull: str = "reference", auto_reset: bool = True):
        if not torch.cuda.is_available():
            raise AuvLibraryError("BatchedAuvEnv needs a GPU (torch.cuda.is_available() is False); "
                                  "there is no CPU fallback")
        self.config = config
        self.device = torch.device(device)
        self.n_envs = int(n_envs)
        self.n_sensors = config.vessel.n_sensors
        # 6 navigation features + closeness per beam (+ 2 velocity channels the reference hard-wires
        # to zero, sensor.py:159) -- config.py:80-98, environment.py:112-114, :263-280
        self.obs_dim = 6 + (config.vessel.n_lidar_observations if config.vessel.use_lidar else 0)
        self._cfg_struct = make_config(config, rewarder=rewarder, test_mode=test_mode, cull=cull,
                                       auto_reset=auto_reset)
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _check(_LIB.auv_create(C.byref(self._cfg_struct), self.n_envs, idx, C.byref(self._h)), "auv_create")
        self._gen = None
        self._fresh = None
        if isinstance(worlds, FreshWorlds):
            # a fresh world on every reset: `depth` slots per environment, regenerated on the device as they are left
            self.fresh_worlds(worlds)
        elif isinstance(worlds, GeneratedWorlds):
            # worlds built on the device from random draws (devgen.py); no host-side bank at all
            self.generate(worlds)
        else:
            if isinstance(worlds, dict):
                bank = worlds
            else:
                bank = pack_bank([w if isinstance(w, BuiltWorld) else build_world(w) for w in worlds])
            self.n_worlds = int(bank["n_worlds"])
            self.k_max = max(1, int(bank["k_max"]))
            self.m_max = max(1, int(bank["m_max"]))
            bs, keep = make_bank_struct(bank)
            _check(_LIB.auv_load_worlds(self._h, C.byref(bs)), "auv_load_worlds")
            del keep
        if auto_reset and self.n_worlds <= self.n_envs and self._fresh is None:
            # auto-reset rebinds env e to world (w + n_envs) % n_worlds: with no more worlds than envs that is the
            # world it has just finished (the reference draws a new scenario on every reset, movingobstacles.py:28-95)
            import warnings
            warnings.warn("BatchedAuvEnv: %d worlds for %d auto-resetting envs -- a finished episode restarts in the SAME world; "
                          "give the bank at least 2 worlds per env, or pass worlds=FreshWorlds() for a new scenario on every reset"
                          % (self.n_worlds, self.n_envs), stacklevel=2)
        # observation_space / action_space exactly as environment.py:101-106, :139-143
        self.action_space = Box(low=np.array([-1, -0.15]), high=np.array([1, 0.15]), dtype=np.float32)
        self.observation_space = Box(low=np.array([-1] * self.obs_dim), high=np.array([1] * self.obs_dim),
                                     dtype=np.float32)
        with torch.cuda.device(self.device):
            self.obs = torch.zeros((self.n_envs, self.obs_dim), dtype=torch.float32, device=self.device)
            self.reward = torch.zeros((self.n_envs,), dtype=torch.float32, device=self.device)
            self.done = torch.zeros((self.n_envs,), dtype=torch.uint8, device=self.device)
        self._graph_actions = None
        self.step_mode = "auto"
        self._slices = None
        self.sub_batches = 1
        # how step_async / step_wait order chains on other streams against the caller's: "device" (one-wave kernels and two
        # words in device memory) beats "cp" (command-processor waits) beats "events" at every chain count measured
        # (4096 x 180, slices all on other streams: 70 / 66 / 43 M env-steps/s with four chains, 109 / 100 / 58 M with two;
        # profiles/r04/sweep_api_final.jsonl)
        self.rendezvous = "device"
        self.stream_probe_s = 0.0
        self._chain_graph = None

    # ------------------------------------------------------------------------------ plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _act(self, actions: torch.Tensor):
        if actions.device != self.device:
            actions = actions.to(self.device)
        if actions.dtype not in (torch.float32, torch.float64):
            actions = actions.to(torch.float32)
        actions = actions.contiguous()
        if tuple(actions.shape) != (self.n_envs, 2):
            raise ValueError("actions must have shape (%d, 2), got %s" % (self.n_envs, tuple(actions.shape)))
        return actions, (_capi.AUV_F64 if actions.dtype == torch.float64 else _capi.AUV_F32)

    def load_worlds(self, worlds: Union[Dict[str, np.ndarray], Sequence[Union[WorldSpec, BuiltWorld]]]):
        """Replace the world bank of this (persistent) handle and put every environment in its reset state:
        `auv_load_worlds` again, no handle tear-down.  What a reset() of the single-environment adapter does."""
        bank = worlds if isinstance(worlds, dict) else pack_bank([w if isinstance(w, BuiltWorld) else build_world(w) for w in worlds])
        torch.cuda.current_stream(self.device).synchronize()
        bs, keep = make_bank_struct(bank)
        _check(_LIB.auv_load_worlds(self._h, C.byref(bs)), "auv_load_worlds")
        del keep
        self._gen = None
        self._fresh = None
        self.n_worlds = int(bank["n_worlds"])
        self.k_max = max(1, int(bank["k_max"]))
        self.m_max = max(1, int(bank["m_max"]))
        self._graph_actions = None
        self._log_first = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _LIB.auv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------ gym-like API
    def reset(self, mask: Optional[torch.Tensor] = None, world_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if m.numel() != self.n_envs:
                raise ValueError("mask must have %d entries, got %d" % (self.n_envs, m.numel()))
        wi = None
        if world_idx is not None and self._fresh is not None:
            raise ValueError("reset(world_idx=...): with FreshWorlds the library binds an environment to its next unseen world")
        if world_idx is not None:
            wi = torch.as_tensor(world_idx).to(device=self.device, dtype=torch.int32).contiguous()
            if wi.numel() != self.n_envs:
                raise ValueError("world_idx must have %d entries, got %d" % (self.n_envs, wi.numel()))
            if int(wi.min()) < 0 or int(wi.max()) >= self.n_worlds:
                # (the library itself ignores an out-of-range entry and keeps that env's binding)
                raise ValueError("world index out of range [0, %d)" % self.n_worlds)
        _check(_LIB.auv_reset(self._h, None if m is None else C.c_void_p(m.data_ptr()),
                              None if wi is None else C.c_void_p(wi.data_ptr()),
                              C.c_void_p(self.obs.data_ptr()), self._stream()), "auv_reset")
        return self.obs

    def step(self, actions: torch.Tensor):
        a, dt = self._act(actions)
        _check(_LIB.auv_step(self._h, C.c_void_p(a.data_ptr()), dt, C.c_void_p(self.obs.data_ptr()),
                             C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr()),
                             self._stream()), "auv_step")
        return self.obs, self.reward, self.done, self._lazy_info()

    # ------------------------------------------------------------------------------ sub-batches
    def _concurrent_streams(self, k: int, first: Optional[torch.cuda.Stream] = None):
        """k streams whose kernels really run side by side.  HIP multiplexes streams onto a few hardware queues (four
        by default, GPU_MAX_HW_QUEUES) and two streams on the same queue serialise, so candidates are tested pairwise
        (auv_streams_overlap: a 300 us do-nothing wave on each) and a set that overlaps mutually is chosen.  Returns
        fewer than k when the device does not offer k (at most four kernels run concurrently on MI355X anyway)."""
        with torch.cuda.device(self.device):
            cands = [torch.cuda.Stream(device=self.device) for _ in range(max(8, 2 * k))]
        ratio = C.c_float()

        def overlap(a, b):
            for _ in range(2):          # (a host hiccup between the two launches reads as "one after the other": ask twice)
                _check(_LIB.auv_streams_overlap(self._h, C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), C.byref(ratio)),
                       "auv_streams_overlap")
                if ratio.value < 1.5:
                    return True
            return False
        if first is not None:
            cands = [first] + cands      # (the caller's stream is chain 0: the others must run side by side with IT)
        chosen = [cands[0]]
        for c in cands[1:]:
            if len(chosen) == k:
                break
            if all(overlap(c, o) for o in chosen):
                chosen.append(c)
        return chosen

    RENDEZVOUS = {"events": _capi.AUV_RDV_EVENTS, "device": _capi.AUV_RDV_DEVICE, "cp": _capi.AUV_RDV_CP}

    def set_sub_batches(self, k: int, probe_streams: bool = True, inline_first: bool = False, strict: bool = False):
        """Split the batch into `k` contiguous sub-batches, each with a stream of its own.  `step_pipelined` /
        `step_async` then step them as k independent launch chains that overlap on the GPU (one sub-batch's sweeps run
        under another's dynamics chain and navigation tail); results are bit-identical to `step`.  k = 1 restores the
        single launch.  Slice boundaries are multiples of 64 environments; if the device runs fewer than k streams
        side by side (see _concurrent_streams) the batch is split into that many -- or, with `strict`, the call raises.
        `inline_first`: the first sub-batch runs on the CALLER's current stream (the one this method is called on) and
        only the others get streams of their own, chosen to run side by side with it: a step_async / step_wait then
        orders k - 1 chains against the caller's stream instead of k."""
        k = int(k)
        if k < 1 or k > 64:
            raise ValueError("sub-batches must be in [1, 64]")
        n = self.n_envs
        torch.cuda.synchronize(self.device)
        cur = torch.cuda.current_stream(self.device)
        t0 = time.perf_counter()
        # (probe_streams = False: any k streams -- under a counter-collecting profiler dispatches are serialised, the
        # probe would find no two streams side by side and the batch would not be split at all)
        # fresh worlds: the refill passes need a stream that shares no hardware queue with a chain -- one MORE stream than chains
        # out of the same mutually-overlapping set (at most four kernels run side by side here: ask for three chains then)
        extra = 1 if (self._fresh is not None and probe_streams) else 0
        if k + extra > 1 and probe_streams:
            # (the probe times two 300 us kernels against the wall clock: a busy host -- eight ranks starting at once -- can
            # make a pair look serialised, so a short selection is tried again before it is believed)
            for _attempt in range(3):
                streams = self._concurrent_streams(k + extra, first=cur if inline_first else None)
                if len(streams) >= k + extra:
                    break
            if extra and len(streams) >= 2:
                self._fresh_stream = streams.pop()            # (kept alive here: the library only borrows it)
                _check(_LIB.auv_fresh_worlds_set_stream(self._h, C.c_void_p(self._fresh_stream.cuda_stream)), "auv_fresh_worlds_set_stream")
        else:
            streams = ([cur] if inline_first else []) + [torch.cuda.Stream(device=self.device) for _ in range(k - int(inline_first))]
        self.stream_probe_s = time.perf_counter() - t0
        if len(streams) < k and strict:
            raise RuntimeError("set_sub_batches(%d): the device runs only %d streams side by side" % (k, len(streams)))
        k = min(k, len(streams))
        per = -(-n // k)
        per = -(-per // 64) * 64
        self._slices = [(lo, min(per, n - lo)) for lo in range(0, n, per)]
        self.sub_batches = len(self._slices)
        self._sub_streams = streams[:self.sub_batches]
        self._bounds_c = (C.c_int32 * (self.sub_batches + 1))(*([lo for lo, _ in self._slices] + [n]))
        self._streams_c = (C.c_void_p * self.sub_batches)(*[st.cuda_stream for st in self._sub_streams])
        self._async_pending = False
        self._chain_graph = None
        # the device-word rendezvous lets a one-wave kernel on one stream poll for a kernel on another: the two must be
        # able to RUN side by side.  Streams that were not measured to (probe_streams = False: e.g. under a counter-
        # collecting profiler, which executes one kernel at a time) get the event-based ordering, which cannot block
        self.rendezvous = "device" if (probe_streams or self.sub_batches == 1) else "events"
        if self.sub_batches > 1 and self.effective_step_mode(per) == "one_launch":
            self.probe_streams()        # the hand-overs' dispatch-order assumption, probed in the shape production runs
        return self._slices

    def step_slice(self, i: int, actions: torch.Tensor, stream: Optional[torch.cuda.Stream] = None):
        """Enqueue one step of sub-batch `i` (see set_sub_batches) on `stream` (default: the sub-batch's own).
        `actions` is the full [N, 2] tensor; only the slice's rows are read, and only the slice's rows of
        obs / reward / done are written.  Ordering against the producer of `actions` is the caller's (stream order,
        events): pass a device tensor of the env's device, float32 / float64, contiguous -- anything else is converted
        on the CURRENT stream and the chain's stream is made to wait for that conversion."""
        lo, cnt = self._slices[i]
        st = self._sub_streams[i] if stream is None else stream
        a, dt = self._act_for(actions, [st])
        _check(_LIB.auv_step_slice(self._h, lo, cnt, C.c_void_p(a.data_ptr()), dt, C.c_void_p(self.obs.data_ptr()),
                                   C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr()),
                                   C.c_void_p(st.cuda_stream)), "auv_step_slice")

    def _act_for(self, actions: torch.Tensor, streams):
        """_act for launches on OTHER streams than the current one.  A conversion (device, dtype, layout) is enqueued on
        the current stream and yields a temporary: the consuming streams must wait for it, and the caching allocator
        must not hand its memory to somebody else while they still read it."""
        a, dt = self._act(actions)
        if a is not actions:
            cur = torch.cuda.current_stream(self.device)
            for st in streams:
                if st != cur:
                    st.wait_stream(cur)
                    a.record_stream(st)
        return a, dt

    def step_pipelined(self, actions: torch.Tensor):
        """One step of the whole batch as `sub_batches` independent launch chains, one C call (auv_step_pipelined):
        sub-batch i goes to its own stream.  Nothing orders the chains against the caller's stream -- for open-loop
        stretches (actions already resident); `step_async` / `step_wait` add that ordering."""
        a, dt = self._act_for(actions, self._sub_streams)
        _check(_LIB.auv_step_pipelined(self._h, self.sub_batches, self._bounds_c, self._streams_c, C.c_void_p(a.data_ptr()), dt,
                                       C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.reward.data_ptr()),
                                       C.c_void_p(self.done.data_ptr())), "auv_step_pipelined")

    def step_multi(self, ring: torch.Tensor, first_slot: int, n_steps: int):
        """`n_steps` consecutive steps of every sub-batch in ONE launch per sub-batch (auv_step_multi): step k reads the actions
        of ring slot (first_slot + k) % slots.  `ring`: [slots, N, 2] float32 / float64, resident on the device.  Open loop, like
        step_pipelined (nothing orders the chains against the caller's stream); bit-identical to n_steps step_pipelined calls."""
        if self._slices is None:
            self.set_sub_batches(1)
        if ring.dim() != 3 or tuple(ring.shape[1:]) != (self.n_envs, 2) or ring.device != self.device or not ring.is_contiguous() \
                or ring.dtype not in (torch.float32, torch.float64):
            raise ValueError("ring must be a contiguous [slots, %d, 2] float32 / float64 tensor on %s" % (self.n_envs, self.device))
        dt = _capi.AUV_F64 if ring.dtype == torch.float64 else _capi.AUV_F32
        _check(_LIB.auv_step_multi(self._h, self.sub_batches, self._bounds_c, self._streams_c, C.c_void_p(ring.data_ptr()), dt,
                                   int(ring.shape[0]), int(first_slot), int(n_steps), C.c_void_p(self.obs.data_ptr()),
                                   C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr())), "auv_step_multi")

    def set_multi_order(self, order: str = "cohorts", lead: int = 16, lag: int = 30):
        """Workgroup order of step_multi's launches: "cohorts" (default: cohort-pipelined, see include/auv_hip.h) or "steps"."""
        _check(_LIB.auv_set_multi_order(self._h, {"steps": 0, "cohorts": 1}[order], int(lead), int(lag)), "auv_set_multi_order")

    def lidar_stage(self, segments: int = 0) -> int:
        """Boundary segments the LiDAR wave stages in LDS per batch of its pair sweep (auv_lidar_stage): picked per bank for the
        one-launch step's occupancy; `segments` 32 .. 96 (even) overrides it.  Results do not depend on it.  Returns the value in force."""
        out = C.c_int32(0)
        _check(_LIB.auv_lidar_stage(self._h, int(segments), C.byref(out)), "auv_lidar_stage")
        return int(out.value)

    def step_pipelined_timed(self, actions: torch.Tensor):
        """step_pipelined with every sub-batch's launch stamped by its own HIP events: ms per sub-batch launch
        (its own duration while the other chains run beside it)."""
        a, dt = self._act_for(actions, self._sub_streams)
        ms = (C.c_float * self.sub_batches)()
        _check(_LIB.auv_step_pipelined_timed(self._h, self.sub_batches, self._bounds_c, self._streams_c, C.c_void_p(a.data_ptr()), dt,
                                             C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.reward.data_ptr()),
                                             C.c_void_p(self.done.data_ptr()), ms), "auv_step_pipelined_timed")
        return [float(x) for x in ms]

    def step_async(self, actions: torch.Tensor, rendezvous: Optional[str] = None):
        """VecEnv.step_async (what SubprocVecEnv does with its workers, scripts/run.py:293-296): enqueue the step of
        every sub-batch on its stream, behind whatever produced `actions` on the caller's current stream, and return at
        once -- ONE C call (auv_step_async); `rendezvous` ("events" | "device" | "cp", default `self.rendezvous`) picks
        how the chains on other streams are ordered against the caller's (include/auv_hip.h, AUV_RDV_*).  The actions
        are converted (device / dtype / layout) FIRST, on the caller's stream, so the chains wait for the conversion
        too, and the converted tensor lives until step_wait."""
        if self._slices is None:
            self.set_sub_batches(1, inline_first=True)
        a, dt = self._act(actions)                       # (on the caller's stream, BEFORE the chains are ordered behind it)
        cur = torch.cuda.current_stream(self.device)
        mode = self.RENDEZVOUS[rendezvous or self.rendezvous]
        _check(_LIB.auv_step_async(self._h, self.sub_batches, self._bounds_c, self._streams_c, C.c_void_p(a.data_ptr()), dt,
                                   C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.reward.data_ptr()),
                                   C.c_void_p(self.done.data_ptr()), C.c_void_p(cur.cuda_stream), mode), "auv_step_async")
        self._async_actions = a                          # the CONVERTED buffer the chains read, alive until step_wait
        self._async_pending = True

    def step_wait(self):
        """VecEnv.step_wait: make the caller's current stream wait for every sub-batch's step (auv_step_wait) and
        return (obs, reward, done, info) -- device tensors, no host synchronisation."""
        if not getattr(self, "_async_pending", False):
            raise RuntimeError("step_wait() without step_async()")
        cur = torch.cuda.current_stream(self.device)
        _check(_LIB.auv_step_wait(self._h, C.c_void_p(cur.cuda_stream)), "auv_step_wait")
        self._async_pending = False
        self._async_actions = None                       # (the caller's stream is ordered behind its last reader now)
        return self.obs, self.reward, self.done, self._lazy_info()

    def set_rendezvous_limit(self, seconds: float):
        """How long a rendezvous kernel of step_async / step_wait ("device") waits before it gives up and reports."""
        _check(_LIB.auv_set_rendezvous_limit(self._h, float(seconds)), "auv_set_rendezvous_limit")

    def _lazy_info(self):
        return _LazyInfo(self)

    STEP_MODES = {"side_by_side": 0, "one_launch": 5, "auto": 6}
    _MODE_NAMES = {v: k for k, v in STEP_MODES.items()}

    def set_step_mode(self, mode: str):
        """"auto" (default): "one_launch" below 65536 environments per launch, "side_by_side" from there on;
        "one_launch": dynamics, LiDAR sweep, navigation search and finish (navigation tail + reward phase) as four roles of
        ONE launch; "side_by_side": K1 ->
        [K2 + K3-nav in one launch] -> K3-reward.  The same bits (tests/test_gpu_parity.py::test_step_modes_agree_bitwise).
        Where the in-launch hand-overs of the first may not be used (see `health()`) the library steps in
        "side_by_side" whatever is set: `effective_step_mode()` tells."""
        _check(_LIB.auv_set_step_mode(self._h, self.STEP_MODES[mode]), "auv_set_step_mode")
        self.step_mode = mode

    def effective_step_mode(self, n_envs_per_launch: int = 0) -> str:
        """The shape a launch of that many environments (0: the whole batch) is really stepped in."""
        return self._MODE_NAMES[_LIB.auv_effective_step_mode(self._h, int(n_envs_per_launch))]

    def health(self) -> Dict[str, int]:
        """State of the in-launch hand-overs (auv_health; reads host memory only, no synchronisation): `handover_ok`,
        `probe_failures` of the last dispatch-order probe, `timeouts` so far, `pending` = a time-out the next step call will
        recover from (the environments the waves that gave up left unfinished are reset, three-launch shape from then on,
        one RuntimeError).  After a recovery `last_timeout()` names the launch that reported and how many were reset.
        A loop that replays a captured step (torch CUDAGraph around `step`) should look at `pending` once per rollout."""
        out = (C.c_int32 * 8)()
        _check(_LIB.auv_health(self._h, out), "auv_health")
        self._health_raw = [int(x) for x in out]
        return dict(handover_ok=int(out[0]), probe_failures=int(out[1]), timeouts=int(out[2]), pending=int(out[3]))

    def rendezvous_state(self) -> Dict[str, int]:
        """`device_ok`: the device-word rendezvous of step_async / step_wait is in use (0: its trial on the chains' streams or a
        real wait has run out once -- the library orders the chains by HIP events from then on, whatever `self.rendezvous`
        says); `timeouts`: how often."""
        self.health()
        return dict(device_ok=int(self._health_raw[7] == 0), timeouts=self._health_raw[7])

    def last_timeout(self) -> Dict[str, int]:
        """The last hand-over time-out this handle recovered from: the slice [e0, e0 + ne) of the launch that reported it
        and the number of environments the recovery reset (e0 = -1: none so far)."""
        self.health()
        return dict(e0=self._health_raw[4], ne=self._health_raw[5], reset_envs=self._health_raw[6])

    def probe_streams(self, streams=None) -> int:
        """Run the dispatch-order probe on the sub-batch streams (auv_probe_streams: one probe launch per stream, all in
        flight together, a foreign kernel behind each); returns the number of polls that ran out (0: the in-launch
        hand-overs stay in use).  set_sub_batches calls it."""
        sts = self._sub_streams if streams is None else streams
        arr = (C.c_void_p * len(sts))(*[st.cuda_stream for st in sts])
        _check(_LIB.auv_probe_streams(self._h, len(sts), arr), "auv_probe_streams")
        return self.health()["probe_failures"]

    # per-kernel entry points (parity tests)
    def step_dynamics(self, actions: torch.Tensor):
        a, dt = self._act(actions)
        _check(_LIB.auv_step_dynamics(self._h, C.c_void_p(a.data_ptr()), dt, self._stream()), "auv_step_dynamics")

    def lidar(self, advance_movers: bool = True):
        _check(_LIB.auv_lidar(self._h, int(advance_movers), self._stream()), "auv_lidar")

    def nav_reward(self, mode: int = 0):
        _check(_LIB.auv_nav_reward(self._h, int(mode), C.c_void_p(self.obs.data_ptr()),
                                   C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr()),
                                   self._stream()), "auv_nav_reward")
        return self.done

    # hipGraph: capture once, replay per step.  `slots` > 1 makes the action buffer a ring of
    # [slots, N, 2]: step k consumes slot k % slots, so nothing has to be copied or re-bound.
    def capture_graph(self, dtype=torch.float32, slots: int = 1, steps: int = 1):
        """Capture `steps` consecutive steps into one hipGraph (replayed by step_graph()).  With `slots` > 1
        the returned [slots, N, 2] tensor is the action ring: replayed step k consumes slot k % slots."""
        self._graph_actions = torch.zeros((slots, self.n_envs, 2), dtype=dtype, device=self.device)
        self._graph_steps = int(steps)
        self._chain_graph = None
        dt = _capi.AUV_F64 if dtype == torch.float64 else _capi.AUV_F32
        torch.cuda.synchronize(self.device)
        _check(_LIB.auv_set_action_ring(self._h, int(slots)), "auv_set_action_ring")
        _check(_LIB.auv_graph_capture_steps(self._h, C.c_void_p(self._graph_actions.data_ptr()), dt,
                                            C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.reward.data_ptr()),
                                            C.c_void_p(self.done.data_ptr()), int(steps), self._stream()), "auv_graph_capture_steps")
        return self._graph_actions if slots > 1 else self._graph_actions[0]

    def capture_graph_chains(self, dtype=torch.float32, slots: int = 1, steps: int = 1, one_graph: bool = False):
        """Capture `steps` consecutive steps of EVERY sub-batch (set_sub_batches) as captured chains (auv_graph_capture_chains):
        by default one linear hipGraph per sub-batch, replayed by step_graph() on the sub-batch's own stream, so that a
        replay keeps `sub_batches` launches in flight like the eager chains do; `one_graph`: one hipGraph with a branch
        per sub-batch, replayed on the caller's stream.  Every chain walks the [slots, N, 2] action ring from slot 0 with
        a position of its own.  Returns the ring."""
        if self._slices is None:
            self.set_sub_batches(1)
        self._graph_actions = torch.zeros((slots, self.n_envs, 2), dtype=dtype, device=self.device)
        self._graph_steps = int(steps)
        dt = _capi.AUV_F64 if dtype == torch.float64 else _capi.AUV_F32
        torch.cuda.synchronize(self.device)
        _check(_LIB.auv_set_action_ring(self._h, int(slots)), "auv_set_action_ring")
        _check(_LIB.auv_graph_capture_chains(self._h, self.sub_batches, self._bounds_c, C.c_void_p(self._graph_actions.data_ptr()), dt,
                                             C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.reward.data_ptr()),
                                             C.c_void_p(self.done.data_ptr()), int(steps), int(bool(one_graph))), "auv_graph_capture_chains")
        self._chain_graph = "one" if one_graph else "per_chain"
        return self._graph_actions if slots > 1 else self._graph_actions[0]

    def step_graph(self, actions: Optional[torch.Tensor] = None):
        if self._graph_actions is None:
            raise RuntimeError("capture_graph() first")
        if actions is not None:
            if self._graph_actions.shape[0] != 1:
                raise ValueError("with an action ring write the slots directly")
            self._graph_actions[0].copy_(actions)
        if self._chain_graph == "per_chain":
            # (nothing orders the chains' streams against the caller's: open-loop stretches, like step_pipelined)
            _check(_LIB.auv_graph_launch_chains(self._h, self.sub_batches, self._streams_c), "auv_graph_launch_chains")
        else:
            _check(_LIB.auv_graph_launch(self._h, self._stream()), "auv_graph_launch")
        return self.obs, self.reward, self.done, self._lazy_info()

    def step_timed(self, actions: torch.Tensor):
        """One step, every dispatch stamped with its own start/stop HIP event; returns four ms values: the launches
        of the effective step mode in order (one_launch: the one launch;
        side_by_side: K1, K2 + K3-nav, K3-reward), zeros, and last the whole step first start .. last stop.
        `timed_kernel_names()` names them."""
        a, dt = self._act(actions)
        ms = (C.c_float * 4)()
        _check(_LIB.auv_step_timed(self._h, C.c_void_p(a.data_ptr()), dt, C.c_void_p(self.obs.data_ptr()),
                                   C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr()),
                                   self._stream(), ms), "auv_step_timed")
        return [float(x) for x in ms]

    def timed_kernel_names(self):
        mode = self.effective_step_mode()
        if mode == "one_launch":
            return ["k_step_roles"]
        return ["k1_dynamics", "k23_lidar_nav", "k3_reward"]

    # ------------------------------------------------------------------------------ optional post-kernel
    def feasibility_pooling(self, width: Optional[float] = None):
        """Sector-wise feasible distances of the current ranges (sensor.py:251-296); returns
        (distances [N, n_sectors] float64, closeness [N, n_sectors] float32)."""
        from .pooling import sector_starts
        v = self.config.vessel
        if width is None:
            width = v.vessel_width * v.feasibility_width_multiplier
        if getattr(self, "_sector_start", None) is None:
            self._sector_start = torch.as_tensor(sector_starts(v.n_sectors, v.n_sensors_per_sector), device=self.device)
        dist = torch.empty((self.n_envs, v.n_sectors), dtype=torch.float64, device=self.device)
        clos = torch.empty((self.n_envs, v.n_sectors), dtype=torch.float32, device=self.device)
        _check(_LIB.auv_feasibility_pooling(self._h, C.c_void_p(self._sector_start.data_ptr()), v.n_sectors,
                                            float(width), C.c_void_p(dist.data_ptr()), C.c_void_p(clos.data_ptr()),
                                            self._stream()), "auv_feasibility_pooling")
        return dist, clos

    # ------------------------------------------------------------------------------ on-device worlds
    def generate(self, spec: "GeneratedWorlds", draws: Optional[torch.Tensor] = None):
        """(Re)build the whole world bank on the device (auv_generate_worlds) and reset every
        environment.  `draws` overrides the spec's seeded draws ([W, n_draws] float64, device)."""
        from . import devgen
        if draws is None:
            draws = devgen.sample_draws(spec.n_worlds, spec.n_moving, spec.n_static, seed=spec.seed, device=self.device)
        draws = draws.to(device=self.device, dtype=torch.float64).contiguous()
        nd = devgen.n_draws(spec.n_moving, spec.n_static)
        if tuple(draws.shape) != (spec.n_worlds, nd):
            raise ValueError("draws must have shape (%d, %d), got %s" % (spec.n_worlds, nd, tuple(draws.shape)))
        unit, nseg = devgen.ring_tables()
        unit = np.ascontiguousarray(unit, dtype=np.float64)
        nseg = np.ascontiguousarray(nseg, dtype=np.int32)
        torch.cuda.current_stream(self.device).synchronize()
        _check(_LIB.auv_generate_worlds(self._h, spec.n_worlds, spec.n_moving, spec.n_static,
                                        C.c_void_p(draws.data_ptr()), nd, unit.ctypes.data_as(C.c_void_p),
                                        nseg.ctypes.data_as(C.c_void_p), len(nseg)), "auv_generate_worlds")
        self._gen = spec
        self._fresh = None
        self._log_first = 0
        self.n_worlds = spec.n_worlds
        self.k_max = max(1, spec.n_moving + spec.n_static)
        self.m_max = max(1, spec.n_moving)
        self._graph_actions = None
        return draws

    def fresh_worlds(self, spec: "FreshWorlds"):
        """Switch to a fresh world on every reset (auv_fresh_worlds_create): builds the bank of depth * n_envs slots on the
        device from the counter-based draws of (seed, global environment index, serial) and resets every environment."""
        from . import devgen
        if not self._cfg_struct.auto_reset:
            raise ValueError("FreshWorlds needs auto_reset=True")
        unit, nseg = devgen.ring_tables()
        unit = np.ascontiguousarray(unit, dtype=np.float64)
        nseg = np.ascontiguousarray(nseg, dtype=np.int32)
        torch.cuda.synchronize(self.device)
        _check(_LIB.auv_fresh_worlds_create(self._h, int(spec.depth), int(spec.n_moving), int(spec.n_static), int(spec.seed),
                                            int(spec.env_index_base), int(spec.batch_cap), int(spec.period),
                                            unit.ctypes.data_as(C.c_void_p), nseg.ctypes.data_as(C.c_void_p), len(nseg)),
               "auv_fresh_worlds_create")
        self._fresh = spec
        # the passes' stream: one that runs side by side with the caller's current stream (set_sub_batches picks again for chains)
        pair = self._concurrent_streams(2, first=torch.cuda.current_stream(self.device))
        if len(pair) == 2:
            self._fresh_stream = pair[1]
            _check(_LIB.auv_fresh_worlds_set_stream(self._h, C.c_void_p(self._fresh_stream.cuda_stream)), "auv_fresh_worlds_set_stream")
        self._gen = GeneratedWorlds(n_worlds=spec.depth * self.n_envs, n_moving=spec.n_moving, n_static=spec.n_static, seed=spec.seed)
        self._log_first = 0
        self.n_worlds = spec.depth * self.n_envs
        self.k_max = max(1, spec.n_moving + spec.n_static)
        self.m_max = max(1, spec.n_moving)
        self._graph_actions = None

    def _chains(self):
        """(n_slices, bounds, streams) of the way the batch is being stepped: the sub-batch chains, or the caller's stream."""
        if self._slices is not None:
            return self.sub_batches, self._bounds_c, self._streams_c
        return 1, (C.c_int32 * 2)(0, self.n_envs), (C.c_void_p * 1)(torch.cuda.current_stream(self.device).cuda_stream)

    def refill(self, flush: bool = False):
        """Fresh worlds: enqueue a refill pass (auv_fresh_worlds_refill).  The step calls of the whole batch do this by themselves
        every `period` calls; a loop that steps slices one by one (step_slice) calls it.  `flush`: synchronise, then run passes
        until every slot left so far is ready again."""
        if self._fresh is None:
            raise RuntimeError("refill(): the env was not built with worlds=FreshWorlds(...)")
        k, b, st = self._chains()
        _check(_LIB.auv_fresh_worlds_refill(self._h, k, b, st, int(bool(flush))), "auv_fresh_worlds_refill")

    def fresh_stats(self) -> Dict[str, int]:
        """`regenerated` worlds rebuilt since the mode was entered, `reused`: episodes that started in the world they had just
        finished because their next slot was not ready (should stay 0), `queued` slots waiting for a pass, `passes_issued` /
        `passes_published` (enqueued / completed), `depth`, `batch_cap`.  One small device-to-host copy."""
        out = (C.c_int64 * 8)()
        _check(_LIB.auv_fresh_worlds_stats(self._h, out), "auv_fresh_worlds_stats")
        return dict(on=int(out[0]), regenerated=int(out[1]), reused=int(out[2]), queued=int(out[3]), passes_issued=int(out[4]),
                    passes_published=int(out[5]), depth=int(out[6]), batch_cap=int(out[7]))

    def fresh_draws(self, envs, serials) -> torch.Tensor:
        """[k, n_draws] float64: the draws of the worlds (environment envs[i] of this handle, serial serials[i]) -- what
        devgen.world_from_draws rebuilds the world from on the host."""
        from . import devgen
        envs = np.ascontiguousarray(envs, dtype=np.int32)
        serials = np.ascontiguousarray(serials, dtype=np.int32)
        nd = devgen.n_draws(self._fresh.n_moving, self._fresh.n_static)
        out = torch.empty((len(envs), nd), dtype=torch.float64, device=self.device)
        _check(_LIB.auv_fresh_worlds_draws(self._h, envs.ctypes.data_as(C.POINTER(C.c_int32)), serials.ctypes.data_as(C.POINTER(C.c_int32)),
                                           len(envs), C.c_void_p(out.data_ptr()), self._stream()), "auv_fresh_worlds_draws")
        return out

    def read_bank(self, name: str) -> torch.Tensor:
        """A table of the generated bank, [W, ...] in slot layout (see _capi.BANK_TABLES)."""
        if self._gen is None:
            raise RuntimeError("read_bank: the bank was uploaded from the host, not generated")
        tid, dtype, tail = _capi.BANK_TABLES[name]
        dims = dict(P=_capi.GEN_POLY_CAP, K=self.k_max, M=self.m_max, G=64 * max(1, self._gen.n_static),
                    C=_capi.GEN_POLY_CAP // 64)
        shape = (self.n_worlds,) + tuple(dims.get(x, x) for x in tail)
        t = torch.empty(shape, dtype=_TORCH_DTYPES[dtype], device=self.device)
        nbytes = t.numel() * t.element_size()
        _check(_LIB.auv_read_bank(self._h, tid, C.c_void_p(t.data_ptr()), nbytes, self._stream()), "auv_read_bank(%s)" % name)
        return t

    # ------------------------------------------------------------------------------ field access
    def field_shape(self, name: str):
        n, S = self.n_envs, self.n_sensors
        return dict(STATE=(6, n), LIDAR_D=(n, S), OBS64=(n, 6 + S), REWARD64=(n,), INFO64=(n, 8),
                    WORLD_IDX=(n,), COUNTERS=(n, 4), MOVER_STATE=(n, self.m_max, 4), NEARBY=(n, self.k_max),
                    EPISODE=(n, 4), CULL_LIMITS=(n, self.k_max, 2), NAV64=(n, 8), COLLISION=(n,), STAMPS=(n, 16), STEP_INFO=(n, 4), BROKEN=(n,),
                    FW_STATE=(self.n_worlds,), FW_SERIAL=(self.n_worlds,))[name]

    def read(self, name: str) -> torch.Tensor:
        t = torch.empty(self.field_shape(name), dtype=_TORCH_DTYPES[FIELD_DTYPES[name]], device=self.device)
        nbytes = t.numel() * t.element_size()
        _check(_LIB.auv_read(self._h, FIELDS[name], C.c_void_p(t.data_ptr()), nbytes, self._stream()),
               "auv_read(%s)" % name)
        return t

    def read_into(self, name: str, dst: torch.Tensor):
        """auv_read straight into `dst` (a contiguous device tensor / slice of exactly the field's size)."""
        nbytes = dst.numel() * dst.element_size()
        _check(_LIB.auv_read(self._h, FIELDS[name], C.c_void_p(dst.data_ptr()), nbytes, self._stream()),
               "auv_read(%s)" % name)
        return dst

    def write(self, name: str, value):
        t = torch.as_tensor(value).to(device=self.device, dtype=_TORCH_DTYPES[FIELD_DTYPES[name]]).contiguous()
        t = t.reshape(self.field_shape(name))
        nbytes = t.numel() * t.element_size()
        _check(_LIB.auv_write(self._h, FIELDS[name], C.c_void_p(t.data_ptr()), nbytes, self._stream()),
               "auv_write(%s)" % name)
        torch.cuda.current_stream(self.device).synchronize()   # `t` may be a temporary

    EPISODE_LOG_COLUMNS = ("env", "reward", "timesteps", "collision", "reached_goal", "progress", "cross_track_error", "world")

    def episode_log(self, max_rows: int = 1 << 20) -> torch.Tensor:
        """Episodes that ended since the last call (or since the bank was loaded), in completion order: [k, 8] float64 on
        the device, columns EPISODE_LOG_COLUMNS -- what the reference appends to `env.history` in save_latest_episode
        (environment.py:466-489), for the whole batch.  One small host synchronisation per call.  The device keeps the
        newest >= max(65536, 4 N) rows: if more episodes than that ended between two calls the oldest are lost, the call
        returns the rows still held and `episode_log_dropped` counts the loss (it never raises for falling behind)."""
        total, start = C.c_int64(), C.c_int64()
        first = getattr(self, "_log_first", 0)
        _check(_LIB.auv_episode_log(self._h, None, 0, first, C.byref(total), C.byref(start), self._stream()), "auv_episode_log")
        first = int(start.value)                                      # (> the cursor when the ring has lapped it)
        k = max(0, min(int(total.value) - first, int(max_rows)))
        rows = torch.empty((k, 8), dtype=torch.float64, device=self.device)
        if k > 0:
            _check(_LIB.auv_episode_log(self._h, C.c_void_p(rows.data_ptr()), k, first, C.byref(total), C.byref(start),
                                        self._stream()), "auv_episode_log")
            first = int(start.value)                                  # (lapped again between the two reads: the k rows copied
                                                                      # start there -- the ring then holds cap >= k of them)
        self.episode_log_dropped = getattr(self, "episode_log_dropped", 0) + (first - getattr(self, "_log_first", 0))
        self._log_first = first + k
        return rows

    def episode_stats(self) -> Dict[str, torch.Tensor]:
        ep = self.read("EPISODE")
        cnt = self.read("COUNTERS")
        return dict(episode_return=ep[:, 0], episode_length=ep[:, 1], collision=ep[:, 2], reached_goal=ep[:, 3],
                    episodes=cnt[:, 2])


class _LazyInfo(dict):
    """info of the batched step (keys as environment.py:336-340): tensors fetched from the
    device only when a key is read.  Values belong to the step that was taken: for an env that
    finished and was auto-reset they are the terminal ones."""
    _KEYS = {"collision": 0, "reached_goal": 1, "goal_distance": 2, "progress": 3}

    def __init__(self, env):
        super().__init__()
        self._env = env
        self._info = None

    def __missing__(self, key):
        if key not in self._KEYS:
            raise KeyError(key)
        if self._info is None:
            self._info = self._env.read("STEP_INFO")
        v = self._info[:, self._KEYS[key]]
        self[key] = v
        return v

    def keys(self):
        return self._KEYS.keys()
