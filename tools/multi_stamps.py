"""Diagnostic (-DAUV_STAMPS_MULTI build): where the wave slots of a multi-step launch go.  One chain, 64 steps per launch; the waves
of the step in the MIDDLE of the last launch leave wall-clock stamps (csrc/k_step_fused.hip); this prints, per role, how long a wave
holds its slot and how much of that is waiting for a hand-over, and the sum over the roles per step against slots x step time.
Usage on the GPU box:
    tools/build_variant.sh mstamps "-DAUV_STAMPS_MULTI"
    AUV_HIP_LIB=gym_auv_amd/csrc_mstamps/libauv_hip.so python tools/multi_stamps.py [workload polygons50|mixed47] [envs] [lead] [lag]"""
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config

wl = sys.argv[1] if len(sys.argv) > 1 else "polygons50"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lead = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lag = int(sys.argv[4]) if len(sys.argv) > 4 else 30
cfg = effective_reference_config(use_lidar=True)
if wl == "mixed47":
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = 16, 16
files = sorted(glob.glob("/tmp/bank.%s.0.%d.*.npz" % (wl, n)))
if not files:
    raise SystemExit("run `python bench.py --workload %s --envs %d --bank-cache /tmp/bank --steps 20` first (it leaves the bank)" % (wl, n))
z = np.load(os.environ.get("BANK") or files[0])
bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
env = BatchedAuvEnv(cfg, bank, n, auto_reset=True)
env.set_sub_batches(1)
env.set_multi_order("cohorts", lead, lag)
env.reset()
g = torch.Generator(device="cuda:0")
g.manual_seed(0)
pool = (torch.rand((64, n, 2), generator=g, device="cuda:0", dtype=torch.float64) * torch.tensor([2.0, 0.3], device="cuda:0", dtype=torch.float64)
        - torch.tensor([1.0, 0.15], device="cuda:0", dtype=torch.float64)).contiguous()
T = 64
for j in range(4):
    env.step_multi(pool, 0, T)
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(8):
    env.step_multi(pool, 0, T)
torch.cuda.synchronize()
us_step = (time.perf_counter() - t0) / (8 * T) * 1e6
st = env.read("STAMPS").cpu().numpy().astype(np.float64)[:, :16] / 100.0      # wall_clock64: 100 MHz -> us


def q(x):
    return dict(mean=round(float(np.mean(x)), 2), p50=round(float(np.percentile(x, 50)), 2), p90=round(float(np.percentile(x, 90)), 2),
                p99=round(float(np.percentile(x, 99)), 2), max=round(float(np.max(x)), 2))


out = dict(workload=wl, envs=n, steps_per_launch=T, lead=lead, lag=lag, lidar_stage=env.lidar_stage(), us_per_step=round(us_step, 2),
           rate_M=round(n / us_step, 1))
sw, se = st[:, 0:6], st[:, 6:10]
dy = st[::1, 10:13]
fi = st[:, 13:16]
roles = {}
roles["sweep"] = dict(waves=n, slot=q(sw[:, 3] - sw[:, 1]), first_instruction_to_role=q(sw[:, 0] - sw[:, 1]), until_state=q(sw[:, 2] - sw[:, 0]),
                      front_and_pairs=q(sw[:, 4] - sw[:, 2]), back_and_publish=q(sw[:, 5] - sw[:, 4]), stores_acknowledged=q(sw[:, 3] - sw[:, 5]))
roles["search"] = dict(waves=n, slot=q(se[:, 3] - se[:, 1]), first_instruction_to_role=q(se[:, 0] - se[:, 1]), until_state=q(se[:, 2] - se[:, 0]),
                       search=q(se[:, 3] - se[:, 2]))
roles["dynamics"] = dict(waves=n // 8, slot=q(dy[:, 2] - dy[:, 0]), wait_carry=q(dy[:, 1] - dy[:, 0]), integrate=q(dy[:, 2] - dy[:, 1]))
roles["finish"] = dict(waves=n // 8, slot=q(fi[:, 2] - fi[:, 0]), until_sweeps_words=q(fi[:, 1] - fi[:, 0]), reward_and_carry=q(fi[:, 2] - fi[:, 1]))
out["roles"] = roles
tot = sum(r["waves"] * r["slot"]["mean"] for r in roles.values())
out["wave_us_per_step"] = round(tot, 0)
out["slot_us_per_step"] = round(4096 * us_step, 0)
out["share_of_slots"] = {k: round(r["waves"] * r["slot"]["mean"] / (4096 * us_step), 3) for k, r in roles.items()}
# the step's own timeline: from the first dynamics wave of the stamped step to its last finish wave
out["step_span_us"] = round(float(fi[:, 2].max() - dy[:, 0].min()), 1)
out["env_critical_path_us"] = q(fi[:, 2] - dy[:, 0])
print(json.dumps(out))
