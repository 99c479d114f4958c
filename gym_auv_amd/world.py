"""WorldSpec -> flat arrays ("world bank") consumed by the HIP library and by the oracle.

Layout = `auv_world_bank_t` in include/auv_hip.h.  Everything here is reset-time host work
(the reference does it inside `_generate()`, envs/movingobstacles.py:28-95); the per-step
path never touches it.
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np

from .obstacles import (circle_segments, enclosing_circle_of_points, polygon_segments)
from .path import Path
from .worldspec import MoverSpec, WorldSpec

OBS_RING, OBS_FILLED, OBS_MOVER = 0, 1, 2
MOVER_NSEG = 5


@dataclass
class BuiltWorld:
    spec: WorldSpec
    path: Path
    scalar: np.ndarray      # [8] L, end_x, end_y, init_x, init_y, init_psi, 0, 0
    obs_meta: np.ndarray    # [K,4] kind, seg_off (world-relative), nseg, mover idx
    obs_cull: np.ndarray    # [K,3]
    seg: np.ndarray         # [G,4]
    mv_param: np.ndarray    # [M,4] width, pos0x, pos0y, n_vel
    mv_init: np.ndarray     # [M,4] px, py, heading, counter
    mv_vtab: List[np.ndarray]


def advance_mover(param: Sequence[float], vtab: np.ndarray, state: Sequence[float], dt: float):
    """VesselObstacle._update (obstacles.py:195-215) for one obstacle; returns new
    (px, py, heading, counter)."""
    _, p0x, p0y, n_vel = param
    px, py, _, counter = state
    counter = counter + dt
    idx = int(math.floor(counter))
    if idx >= int(n_vel) - 1:
        counter, idx = 0.0, 0
        px, py = p0x, p0y
    vx, vy = vtab[min(idx, len(vtab) - 1)]
    dx, dy = dt * vx, dt * vy
    return (px + dx, py + dy, math.atan2(dy, dx), counter)


def build_world(spec: WorldSpec) -> BuiltWorld:
    path = Path(spec.waypoints)
    end = path.end
    scalar = np.array([path.length, end[0], end[1], spec.vessel_init[0], spec.vessel_init[1],
                       spec.vessel_init[2], 0.0, 0.0])
    meta, cull, segs = [], [], []
    off = 0
    for cx, cy, r in np.asarray(spec.circles, dtype=np.float64).reshape(-1, 3):
        s = circle_segments(cx, cy, r)
        meta.append([OBS_RING, off, len(s), -1])
        cull.append([cx, cy, r])
        segs.append(s)
        off += len(s)
    for pts in spec.polygons:
        s = polygon_segments(pts)
        meta.append([OBS_FILLED, off, len(s), -1])
        cull.append(list(enclosing_circle_of_points(pts)))
        segs.append(s)
        off += len(s)
    mv_param, mv_init, mv_vtab = [], [], []
    for k, m in enumerate(spec.movers):
        meta.append([OBS_MOVER, 0, MOVER_NSEG, k])
        cull.append([0.0, 0.0, 0.0])
        mv_param.append([m.width, m.pos0[0], m.pos0[1], float(m.n_vel)])
        mv_init.append([m.pos[0], m.pos[1], m.heading, m.counter])
        mv_vtab.append(np.asarray(m.vel, dtype=np.float64).reshape(-1, 2))
    return BuiltWorld(
        spec=spec, path=path, scalar=scalar,
        obs_meta=np.asarray(meta, dtype=np.int32).reshape(-1, 4),
        obs_cull=np.asarray(cull, dtype=np.float64).reshape(-1, 3),
        seg=(np.concatenate(segs) if segs else np.zeros((0, 4))),
        mv_param=np.asarray(mv_param, dtype=np.float64).reshape(-1, 4),
        mv_init=np.asarray(mv_init, dtype=np.float64).reshape(-1, 4),
        mv_vtab=mv_vtab)


def pack_bank(worlds: Sequence[BuiltWorld]) -> Dict[str, np.ndarray]:
    """Concatenate built worlds into the CSR arrays of auv_world_bank_t (C-contiguous)."""
    W = len(worlds)
    poly_off = np.zeros(W + 1, dtype=np.int64)
    knot_off = np.zeros(W + 1, dtype=np.int64)
    obs_off = np.zeros(W + 1, dtype=np.int64)
    mv_off = np.zeros(W + 1, dtype=np.int64)
    poly_xy, poly_cum, knot_s, knot_coef, scal = [], [], [], [], []
    obs_meta, obs_cull, seg, mv_param, mv_init, vtabs = [], [], [], [], [], []
    seg_base = 0
    for i, w in enumerate(worlds):
        p = w.path
        poly_off[i + 1] = poly_off[i] + len(p.points)
        knot_off[i + 1] = knot_off[i] + len(p.knot_s)
        obs_off[i + 1] = obs_off[i] + len(w.obs_meta)
        mv_off[i + 1] = mv_off[i] + len(w.mv_param)
        poly_xy.append(p.points)
        poly_cum.append(p._cum)
        knot_s.append(p.knot_s)
        coef = np.zeros((len(p.knot_s), 8))
        coef[:-1, 0:4] = p.cx.T
        coef[:-1, 4:8] = p.cy.T
        knot_coef.append(coef)
        scal.append(w.scalar)
        m = w.obs_meta.copy()
        m[:, 1] += np.where(m[:, 0] == OBS_MOVER, 0, seg_base).astype(np.int32)
        obs_meta.append(m)
        obs_cull.append(w.obs_cull)
        seg.append(w.seg)
        seg_base += len(w.seg)
        mv_param.append(w.mv_param)
        mv_init.append(w.mv_init)
        vtabs.extend(w.mv_vtab)
    vt_off = np.zeros(len(vtabs) + 1, dtype=np.int64)
    for i, v in enumerate(vtabs):
        vt_off[i + 1] = vt_off[i] + len(v)

    def cat(xs, shape, dtype=np.float64):
        xs = [np.asarray(x, dtype=dtype).reshape(shape) for x in xs]
        empty = np.zeros((0,) + tuple(shape[1:]), dtype=dtype)
        return np.ascontiguousarray(np.concatenate(xs + [empty]))

    bank = dict(
        n_worlds=np.int32(W),
        poly_off=poly_off, poly_xy=cat(poly_xy, (-1, 2)), poly_cum=cat(poly_cum, (-1,)),
        knot_off=knot_off, knot_s=cat(knot_s, (-1,)), knot_coef=cat(knot_coef, (-1, 8)),
        world_scalar=cat(scal, (-1, 8)),
        obs_off=obs_off, obs_meta=cat(obs_meta, (-1, 4), np.int32), obs_cull=cat(obs_cull, (-1, 3)),
        seg=cat(seg, (-1, 4)),
        mv_off=mv_off, mv_param=cat(mv_param, (-1, 4)), mv_init=cat(mv_init, (-1, 4)),
        mv_vtab_off=vt_off, mv_vtab=cat(vtabs, (-1, 2)),
    )
    bank["k_max"] = int(np.diff(obs_off).max()) if W else 0
    bank["m_max"] = int(np.diff(mv_off).max()) if W else 0
    bank["p_max"] = int(np.diff(poly_off).max()) if W else 0
    return bank


def merge_banks(banks: Sequence[Dict[str, np.ndarray]]) -> Dict[str, np.ndarray]:
    """Concatenate packed banks (e.g. built by parallel workers) into one, fixing offsets."""
    out: Dict[str, np.ndarray] = {}

    def cat_off(name):
        parts, base = [np.zeros(1, dtype=np.int64)], 0
        for b in banks:
            parts.append(b[name][1:] + base)
            base += int(b[name][-1])
        return np.concatenate(parts)

    for name in ("poly_off", "knot_off", "obs_off", "mv_off", "mv_vtab_off"):
        out[name] = cat_off(name)
    for name in ("poly_xy", "poly_cum", "knot_s", "knot_coef", "world_scalar", "obs_cull", "seg",
                 "mv_param", "mv_init", "mv_vtab"):
        out[name] = np.ascontiguousarray(np.concatenate([b[name] for b in banks]))
    metas, seg_base = [], 0
    for b in banks:
        m = b["obs_meta"].copy()
        m[:, 1] += np.where(m[:, 0] == OBS_MOVER, 0, seg_base).astype(np.int32)
        metas.append(m)
        seg_base += len(b["seg"])
    out["obs_meta"] = np.ascontiguousarray(np.concatenate(metas))
    out["n_worlds"] = np.int32(sum(int(b["n_worlds"]) for b in banks))
    out["k_max"] = max(int(b["k_max"]) for b in banks)
    out["m_max"] = max(int(b["m_max"]) for b in banks)
    out["p_max"] = max(int(b["p_max"]) for b in banks)
    return out


def _build_chunk(args):
    kind, seeds, kwargs = args
    from . import scenarios
    gen = getattr(scenarios, kind)
    return pack_bank([build_world(gen(int(s), **kwargs)) for s in seeds])


def build_bank_parallel(kind: str, seeds: Sequence[int], procs: int = 1, **kwargs) -> Dict[str, np.ndarray]:
    """Generate + build + pack worlds `scenarios.<kind>(seed, **kwargs)` for all seeds using
    `procs` worker processes (reset-time host work; the reference does this serially in
    `_generate()`)."""
    seeds = list(seeds)
    if procs <= 1 or len(seeds) < 2 * procs:
        return _build_chunk((kind, seeds, kwargs))
    import multiprocessing as mp
    # CONTIGUOUS chunks merged in order: world i of the bank is always seeds[i], whatever `procs` is
    # (so env e -> world e does not depend on the core or rank count)
    n_chunks = min(len(seeds), procs * 4)
    bounds = [len(seeds) * c // n_chunks for c in range(n_chunks + 1)]
    chunks = [(kind, seeds[bounds[c]:bounds[c + 1]], kwargs) for c in range(n_chunks)]
    with mp.get_context("fork").Pool(procs) as pool:
        banks = pool.map(_build_chunk, chunks)
    return merge_banks(banks)
