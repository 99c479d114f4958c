"""ctypes wrapper of oracle/_build/libauv_oracle.so (auv_oracle.c).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package gym_auv_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from gym_auv_amd._capi import (AuvConfig, AuvWorldBank, FIELDS, FIELD_DTYPES, make_bank_struct)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libauv_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "auv_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "auv_hip.h")
    if (force or not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def _load():
    global _lib
    if _lib is None:
        # AUV_ORACLE_LIB points the tests at another build of the same source (e.g. `make asan`)
        lib = C.CDLL(os.environ.get("AUV_ORACLE_LIB") or build())
        vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
        lib.oracle_create.argtypes = [C.POINTER(AuvConfig), i32, C.POINTER(vp)]
        lib.oracle_destroy.argtypes = [vp]
        lib.oracle_load_worlds.argtypes = [vp, C.POINTER(AuvWorldBank)]
        lib.oracle_reset.argtypes = [vp, vp, vp]
        lib.oracle_step.argtypes = [vp, vp, vp]
        lib.oracle_step_dynamics.argtypes = [vp, vp]
        lib.oracle_lidar.argtypes = [vp, i32]
        lib.oracle_nav_reward.argtypes = [vp, i32, vp]
        lib.oracle_read.argtypes = [vp, i32, vp, sz]
        lib.oracle_write.argtypes = [vp, i32, vp, sz]
        lib.oracle_field_bytes.argtypes = [vp, i32]
        lib.oracle_field_bytes.restype = sz
        lib.oracle_kmax.argtypes = [vp]
        lib.oracle_set_threads.argtypes = [C.c_int]
        lib.oracle_feasibility_pooling.argtypes = [vp, vp, i32, C.c_double, vp]
        lib.oracle_mmax.argtypes = [vp]
        _lib = lib
    return _lib


class Oracle:
    """Batched CPU oracle with the same call sequence as the HIP handle."""

    def __init__(self, cfg_struct: AuvConfig, n_envs: int, bank: dict):
        self.lib = _load()
        self.cfg = cfg_struct
        self.n = n_envs
        self.S = cfg_struct.n_sensors
        self.h = C.c_void_p()
        assert self.lib.oracle_create(C.byref(cfg_struct), n_envs, C.byref(self.h)) == 0
        self._bank_struct, self._keep = make_bank_struct(bank)
        assert self.lib.oracle_load_worlds(self.h, C.byref(self._bank_struct)) == 0
        self.k_max = self.lib.oracle_kmax(self.h)
        self.m_max = self.lib.oracle_mmax(self.h)

    def close(self):
        if self.h:
            self.lib.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- calls -----------------------------------------------------------------------
    def reset(self, mask=None, world_idx=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        w = None if world_idx is None else np.ascontiguousarray(world_idx, dtype=np.int32)
        rc = self.lib.oracle_reset(self.h, None if m is None else m.ctypes.data,
                                   None if w is None else w.ctypes.data)
        assert rc == 0, rc
        return self.read("OBS64")

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.n, 2)
        done = np.zeros(self.n, dtype=np.uint8)
        assert self.lib.oracle_step(self.h, a.ctypes.data, done.ctypes.data) == 0
        return self.read("OBS64"), self.read("REWARD64"), done

    def step_dynamics(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.n, 2)
        assert self.lib.oracle_step_dynamics(self.h, a.ctypes.data) == 0

    def lidar(self, advance_movers=True):
        assert self.lib.oracle_lidar(self.h, int(advance_movers)) == 0

    def nav_reward(self, mode=0):
        done = np.zeros(self.n, dtype=np.uint8)
        assert self.lib.oracle_nav_reward(self.h, int(mode), done.ctypes.data) == 0
        return done

    def feasibility_pooling(self, sector_start, width):
        st = np.ascontiguousarray(sector_start, dtype=np.int32)
        out = np.empty((self.n, len(st) - 1), dtype=np.float64)
        assert self.lib.oracle_feasibility_pooling(self.h, st.ctypes.data, len(st) - 1, float(width), out.ctypes.data) == 0
        return out

    # --- fields ------------------------------------------------------------------------
    def _shape(self, name):
        n, S = self.n, self.S
        return dict(STATE=(6, n), LIDAR_D=(n, S), OBS64=(n, 6 + S), REWARD64=(n,), INFO64=(n, 8),
                    WORLD_IDX=(n,), COUNTERS=(n, 4), MOVER_STATE=(n, self.m_max, 4),
                    NEARBY=(n, self.k_max), EPISODE=(n, 4), CULL_LIMITS=(n, self.k_max, 2),
                    NAV64=(n, 8), COLLISION=(n,), STEP_INFO=(n, 4))[name]

    def read(self, name):
        out = np.empty(self._shape(name), dtype=FIELD_DTYPES[name])
        rc = self.lib.oracle_read(self.h, FIELDS[name], out.ctypes.data, out.nbytes)
        assert rc == 0, (name, rc, out.nbytes, self.lib.oracle_field_bytes(self.h, FIELDS[name]))
        return out

    def write(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=FIELD_DTYPES[name]).reshape(self._shape(name))
        assert self.lib.oracle_write(self.h, FIELDS[name], a.ctypes.data, a.nbytes) == 0


def set_threads(n: int) -> int:
    """Set (n > 0) / query the OpenMP thread count of the oracle's batched loops."""
    return int(_load().oracle_set_threads(int(n)))
