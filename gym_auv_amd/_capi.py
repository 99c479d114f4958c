"""ctypes binding of include/auv_hip.h (the C ABI of libauv_hip.so).

The library is REQUIRED: there is no CPU or PyTorch fallback in the product path.  If the
shared object is missing or does not export the expected ABI, importing the batched env
raises immediately (`AuvLibraryError`).
"""
import ctypes as C
import os
from typing import Dict, Tuple

import numpy as np

from .config import Config

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libauv_hip.so")
ABI_VERSION = 4

AUV_REWARD_COLAV, AUV_REWARD_PATHFOLLOW = 0, 1
AUV_CULL_REFERENCE, AUV_CULL_EXACT = 0, 1
AUV_F32, AUV_F64 = 0, 1
AUV_RDV_EVENTS, AUV_RDV_DEVICE, AUV_RDV_CP = 0, 1, 2

FIELDS = dict(STATE=0, LIDAR_D=1, OBS64=2, REWARD64=3, INFO64=4, WORLD_IDX=5, COUNTERS=6,
              MOVER_STATE=7, NEARBY=8, EPISODE=9, CULL_LIMITS=10, NAV64=11, COLLISION=12, STAMPS=13, STEP_INFO=14, BROKEN=15, FW_STATE=16, FW_SERIAL=17)
FIELD_DTYPES = dict(STATE=np.float64, LIDAR_D=np.float64, OBS64=np.float64, REWARD64=np.float64,
                    INFO64=np.float64, WORLD_IDX=np.int32, COUNTERS=np.int32, MOVER_STATE=np.float64,
                    NEARBY=np.uint8, EPISODE=np.float64, CULL_LIMITS=np.int32, NAV64=np.float64,
                    COLLISION=np.uint8, STAMPS=np.int64, STEP_INFO=np.float64, BROKEN=np.uint8,
                    FW_STATE=np.int32, FW_SERIAL=np.int32)


class AuvLibraryError(RuntimeError):
    pass


class AuvConfig(C.Structure):
    _fields_ = [
        ("dt", C.c_double), ("min_goal_distance", C.c_double), ("min_path_progress", C.c_double),
        ("min_cumulative_reward", C.c_double), ("sensor_range", C.c_double),
        ("vessel_width", C.c_double), ("look_ahead_distance", C.c_double),
        ("thrust_max", C.c_double), ("moment_max", C.c_double),
        ("max_timesteps", C.c_int32), ("n_sensors", C.c_int32),
        ("sensor_interval_load_obstacles", C.c_int32), ("use_lidar", C.c_int32),
        ("sensor_log_transform", C.c_int32), ("rewarder", C.c_int32), ("test_mode", C.c_int32),
        ("cull_mode", C.c_int32), ("auto_reset", C.c_int32), ("obs_channels", C.c_int32),
    ]


_I64P, _I32P, _F64P = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)


class AuvWorldBank(C.Structure):
    _fields_ = [
        ("n_worlds", C.c_int32),
        ("poly_off", _I64P), ("poly_xy", _F64P), ("poly_cum", _F64P),
        ("knot_off", _I64P), ("knot_s", _F64P), ("knot_coef", _F64P), ("world_scalar", _F64P),
        ("obs_off", _I64P), ("obs_meta", _I32P), ("obs_cull", _F64P),
        ("n_seg", C.c_int64), ("seg", _F64P),
        ("mv_off", _I64P), ("mv_param", _F64P), ("mv_init", _F64P),
        ("mv_vtab_off", _I64P), ("mv_vtab", _F64P),
    ]


def make_config(cfg: Config, rewarder: str = "colav", test_mode: bool = False,
                cull: str = "reference", auto_reset: bool = False) -> AuvConfig:
    v, e, s = cfg.vessel, cfg.episode, cfg.simulation
    if v.sensor_use_feasibility_pooling:
        raise NotImplementedError("feasibility pooling is not part of step() (its wiring is broken in the reference, "
                                  "SURVEY 8(f) F3); use BatchedAuvEnv.feasibility_pooling() on the ranges")
    if v.sensor_use_velocity_observations and not v.use_lidar:
        raise ValueError("sensor_use_velocity_observations needs use_lidar (the reference raises on [].flatten(), "
                         "environment.py:260,271-272)")
    return AuvConfig(
        dt=s.t_step_size, min_goal_distance=e.min_goal_distance, min_path_progress=e.min_path_progress,
        min_cumulative_reward=e.min_cumulative_reward, sensor_range=v.sensor_range,
        vessel_width=v.vessel_width, look_ahead_distance=float(v.look_ahead_distance),
        thrust_max=v.thrust_max_auv, moment_max=v.moment_max_auv, max_timesteps=e.max_timesteps,
        n_sensors=v.n_sensors, sensor_interval_load_obstacles=v.sensor_interval_load_obstacles,
        use_lidar=int(bool(v.use_lidar)), sensor_log_transform=int(bool(v.sensor_log_transform)),
        rewarder={"colav": AUV_REWARD_COLAV, "pathfollow": AUV_REWARD_PATHFOLLOW}[rewarder],
        test_mode=int(test_mode), cull_mode={"reference": AUV_CULL_REFERENCE, "exact": AUV_CULL_EXACT}[cull],
        auto_reset=int(auto_reset), obs_channels=3 if v.sensor_use_velocity_observations else 1)


def make_bank_struct(bank: Dict[str, np.ndarray]) -> Tuple[AuvWorldBank, list]:
    """Wrap the numpy arrays of world.pack_bank() without copying; returns (struct, keepalive)."""
    keep = []

    def ptr(name, dtype, ctype):
        a = np.ascontiguousarray(bank[name], dtype=dtype)
        if a.size == 0:
            a = np.zeros(8, dtype=dtype)   # never hand a NULL/dangling pointer across the ABI
        keep.append(a)
        return a.ctypes.data_as(ctype)

    s = AuvWorldBank()
    s.n_worlds = int(bank["n_worlds"])
    for name in ("poly_off", "knot_off", "obs_off", "mv_off", "mv_vtab_off"):
        setattr(s, name, ptr(name, np.int64, _I64P))
    for name in ("poly_xy", "poly_cum", "knot_s", "knot_coef", "world_scalar", "obs_cull", "seg",
                 "mv_param", "mv_init", "mv_vtab"):
        setattr(s, name, ptr(name, np.float64, _F64P))
    s.obs_meta = ptr("obs_meta", np.int32, _I32P)
    s.n_seg = int(len(bank["seg"]))
    return s, keep


class AuvPolicyIO(C.Structure):
    """== auv_policy_io_t (include/auv_hip.h): the buffers of one sub-batch's policy launch"""
    _fields_ = [
        ("obs", C.c_void_p), ("params", C.c_void_p), ("ctr", C.c_void_p), ("reward_in", C.c_void_p), ("done_in", C.c_void_p),
        ("actions_out", C.c_void_p), ("O", C.c_void_p), ("A", C.c_void_p), ("LP", C.c_void_p), ("V", C.c_void_p),
        ("R", C.c_void_p), ("Dn", C.c_void_p), ("mu_out", C.c_void_p), ("eps_out", C.c_void_p),
        ("seed", C.c_uint64), ("obs_dim", C.c_int32), ("T", C.c_int32), ("ld", C.c_int32), ("env_base", C.c_int32),
        ("act_mid", C.c_float * 2), ("act_half", C.c_float * 2), ("clip_lo", C.c_float * 2), ("clip_hi", C.c_float * 2),
        ("reward_scale", C.c_float), ("reward_clip", C.c_float), ("params_bf16", C.c_void_p),
    ]


_lib = None


HOOKS_LIB_PATH = os.path.join(_HERE, "csrc", "libauv_hip_hooks.so")   # `make -C gym_auv_amd/csrc hooks`: tests only


def load_library(path: str = None) -> C.CDLL:
    """dlopen libauv_hip.so and declare every prototype of include/auv_hip.h.
    AUV_HIP_LIB selects another build of the same library (A/B timing of kernel variants; the test-hook build)."""
    global _lib
    if _lib is not None:
        return _lib
    if path is None:
        path = os.environ.get("AUV_HIP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise AuvLibraryError(
            "HIP extension not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C gym_auv_amd/csrc`). There is no CPU fallback." % path)
    try:
        lib = C.CDLL(path)
    except OSError as exc:  # e.g. libamdhip64 not found
        raise AuvLibraryError("cannot load %s: %s" % (path, exc)) from exc
    vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
    protos = {
        "auv_create": (C.c_int, [C.POINTER(AuvConfig), i32, i32, C.POINTER(vp)]),
        "auv_destroy": (C.c_int, [vp]),
        "auv_load_worlds": (C.c_int, [vp, C.POINTER(AuvWorldBank)]),
        "auv_reset": (C.c_int, [vp, vp, vp, vp, vp]),
        "auv_step": (C.c_int, [vp, vp, i32, vp, vp, vp, vp]),
        "auv_step_slice": (C.c_int, [vp, i32, i32, vp, i32, vp, vp, vp, vp]),
        "auv_step_pipelined": (C.c_int, [vp, i32, C.POINTER(i32), C.POINTER(vp), vp, i32, vp, vp, vp]),
        "auv_step_multi": (C.c_int, [vp, i32, C.POINTER(i32), C.POINTER(vp), vp, i32, i32, i32, i32, vp, vp, vp]),
        "auv_set_multi_order": (C.c_int, [vp, i32, i32, i32]),
        "auv_lidar_stage": (C.c_int, [vp, i32, C.POINTER(i32)]),
        "auv_step_async": (C.c_int, [vp, i32, C.POINTER(i32), C.POINTER(vp), vp, i32, vp, vp, vp, vp, i32]),
        "auv_step_wait": (C.c_int, [vp, vp]),
        "auv_set_rendezvous_limit": (C.c_int, [vp, C.c_double]),
        "auv_graph_capture_chains": (C.c_int, [vp, i32, C.POINTER(i32), vp, i32, vp, vp, vp, i32, i32]),
        "auv_graph_launch_chains": (C.c_int, [vp, i32, C.POINTER(vp)]),
        "auv_policy_param_floats": (sz, [i32]),
        "auv_policy_act": (C.c_int, [vp, i32, i32, C.POINTER(AuvPolicyIO), vp]),
        "auv_gae": (C.c_int, [vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, i32, i32, vp]),
        "auv_policy_rollout": (C.c_int, [vp, i32, C.POINTER(i32), C.POINTER(vp), C.POINTER(AuvPolicyIO), vp, vp, vp, i32, i32,
                                         C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "auv_step_pipelined_timed": (C.c_int, [vp, i32, C.POINTER(i32), C.POINTER(vp), vp, i32, vp, vp, vp, C.POINTER(C.c_float)]),
        "auv_streams_overlap": (C.c_int, [vp, vp, vp, C.POINTER(C.c_float)]),
        "auv_episode_log": (C.c_int, [vp, vp, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp]),
        "auv_health": (C.c_int, [vp, C.POINTER(i32)]),
        "auv_probe_streams": (C.c_int, [vp, i32, C.POINTER(vp)]),
        "auv_effective_step_mode": (C.c_int, [vp, i32]),
        "auv_step_dynamics": (C.c_int, [vp, vp, i32, vp]),
        "auv_lidar": (C.c_int, [vp, i32, vp]),
        "auv_nav_reward": (C.c_int, [vp, i32, vp, vp, vp, vp]),
        "auv_read": (C.c_int, [vp, i32, vp, sz, vp]),
        "auv_write": (C.c_int, [vp, i32, vp, sz, vp]),
        "auv_field_bytes": (sz, [vp, i32]),
        "auv_graph_capture": (C.c_int, [vp, vp, i32, vp, vp, vp, vp]),
        "auv_graph_launch": (C.c_int, [vp, vp]),
        "auv_graph_capture_steps": (C.c_int, [vp, vp, i32, vp, vp, vp, i32, vp]),
        "auv_step_timed": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, C.POINTER(C.c_float)]),
        "auv_set_action_ring": (C.c_int, [vp, i32]),
        "auv_set_step_mode": (C.c_int, [vp, i32]),
        "auv_feasibility_pooling": (C.c_int, [vp, vp, i32, C.c_double, vp, vp, vp]),
        "auv_generate_worlds": (C.c_int, [vp, i32, i32, i32, vp, i32, vp, vp, i32]),
        "auv_fresh_worlds_create": (C.c_int, [vp, i32, i32, i32, C.c_uint64, C.c_int64, i32, i32, vp, vp, i32]),
        "auv_fresh_worlds_refill": (C.c_int, [vp, i32, C.POINTER(i32), C.POINTER(vp), i32]),
        "auv_fresh_worlds_stats": (C.c_int, [vp, C.POINTER(C.c_int64)]),
        "auv_fresh_worlds_set_stream": (C.c_int, [vp, vp]),
        "auv_fresh_worlds_draws": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), i32, vp, vp]),
        "auv_bank_bytes": (sz, [vp, i32]),
        "auv_read_bank": (C.c_int, [vp, i32, vp, sz, vp]),
        "auv_abi_version": (i32, []),
        "auv_last_error": (C.c_char_p, []),
    }
    for name, (res, args) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise AuvLibraryError("%s does not export %s" % (path, name)) from exc
        fn.restype, fn.argtypes = res, args
    if lib.auv_abi_version() != ABI_VERSION:
        raise AuvLibraryError("ABI mismatch: library %d, binding %d" % (lib.auv_abi_version(), ABI_VERSION))
    if hasattr(lib, "auv_test_hooks"):   # only the -DAUV_TEST_HOOKS build exports it
        lib.auv_test_hooks.restype, lib.auv_test_hooks.argtypes = C.c_int, [vp, i32, i32]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = ["auv_create", "auv_destroy", "auv_load_worlds", "auv_reset", "auv_step", "auv_step_slice",
                    "auv_step_pipelined", "auv_step_multi", "auv_set_multi_order", "auv_lidar_stage", "auv_step_async", "auv_step_wait", "auv_set_rendezvous_limit", "auv_graph_capture_chains",
                    "auv_graph_launch_chains", "auv_policy_param_floats", "auv_policy_act", "auv_gae", "auv_policy_rollout", "auv_step_pipelined_timed", "auv_streams_overlap", "auv_episode_log", "auv_health", "auv_probe_streams", "auv_effective_step_mode",
                    "auv_step_dynamics", "auv_lidar", "auv_nav_reward", "auv_read", "auv_write",
                    "auv_field_bytes", "auv_graph_capture", "auv_graph_launch", "auv_graph_capture_steps", "auv_step_timed",
                    "auv_set_action_ring", "auv_set_step_mode", "auv_feasibility_pooling",
                    "auv_generate_worlds", "auv_bank_bytes", "auv_read_bank",
                    "auv_fresh_worlds_create", "auv_fresh_worlds_refill", "auv_fresh_worlds_stats", "auv_fresh_worlds_draws", "auv_fresh_worlds_set_stream",
                    "auv_abi_version", "auv_last_error"]

# tables of a generated bank (auv_read_bank): id, dtype, trailing shape ('P' = AUV_GEN_POLY_CAP,
# 'K' / 'M' / 'G' = obstacles / movers / segment slots per world)
GEN_POLY_CAP = 16384
GEN_CAND = 8
BANK_TABLES = {
    "POLY_CNT": (0, np.int32, ()), "POLY_XY": (1, np.float64, ("P", 2)), "POLY_CUM": (2, np.float64, ("P",)),
    "KNOT_S": (3, np.float64, (1000,)), "KNOT_COEF": (4, np.float64, (1000, 8)), "WORLD_SCALAR": (5, np.float64, (8,)),
    "OBS_META": (6, np.int32, ("K", 4)), "OBS_CULL": (7, np.float64, ("K", 3)), "SEG": (8, np.float64, ("G", 4)),
    "MV_PARAM": (9, np.float64, ("M", 4)), "MV_INIT": (10, np.float64, ("M", 4)), "MV_VTAB": (11, np.float64, ("M", 2)),
    "CHUNK_BOUND": (12, np.float64, ("C", 4)),
}
