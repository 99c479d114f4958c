"""Diagnostic (STAMPS build): wave timelines of the one-launch step (k_step_roles), for one chain over the whole batch or
for K sub-batch chains running side by side -- per chain, relative to the chain's own first wave start of its last step.
Usage on the GPU box:
    tools/build_variant.sh stamps "-DAUV_STAMPS"
    AUV_HIP_LIB=gym_auv_amd/csrc_stamps/libauv_hip.so SUB=4 python tools/phase_stamps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config

cfg = effective_reference_config(use_lidar=True)
n = 4096
z = np.load(os.environ.get("BANK") or sorted(__import__("glob").glob("/tmp/bank.polygons50.0.4096.4096.2*.npz"))[0])   # (bench.py --bank-cache /tmp/bank: name carries a source hash)
bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
env = BatchedAuvEnv(cfg, bank, n, auto_reset=True)
env.set_step_mode("one_launch")
sub = int(os.environ.get("SUB", "4"))
slices = env.set_sub_batches(sub)
env.reset()
g = torch.Generator(device="cuda:0")
g.manual_seed(0)
pool = torch.rand((64, n, 2), generator=g, device="cuda:0") * torch.tensor([2.0, 0.3], device="cuda:0") - torch.tensor([1.0, 0.15], device="cuda:0")
for i in range(int(os.environ.get("STEPS", "400"))):
    if env.sub_batches > 1:
        env.step_pipelined(pool[i % 64])
    else:
        env.step(pool[i % 64])
torch.cuda.synchronize()
st = env.read("STAMPS").cpu().numpy().astype(np.float64)


def f(nm, x, q=(50, 90, 99, 100)):
    print("  %-46s" % nm, " ".join("p%d %6.2f" % (qq, v) for qq, v in zip(q, np.percentile(x / 100.0, list(q)))))


for ci, (lo, cnt) in enumerate(env._slices):
    s = st[lo:lo + cnt]
    k1, l0, lp, ls, l1, l2 = s[:, 0], s[:, 3], s[:, 1], s[:, 2], s[:, 4], s[:, 14]
    n0, ns, n1, n2 = s[:, 12], s[:, 5], s[:, 13], s[:, 15]
    base = l0.min()
    print("chain %d: envs [%d, %d), launch span %.1f us (first LiDAR wave start .. last environment finished)" % (ci, lo, lo + cnt, (n2.max() - base) / 100.0))
    f("dynamics: state published (us)", k1 - base)
    f("LiDAR start offsets (us)", l0 - base)
    f("LiDAR pre-work done (us)", lp - base)
    f("LiDAR state arrived (us)", ls - base)
    f("LiDAR sweep duration from state (us)", l1 - ls)
    f("LiDAR sweep end (us)", l1 - base)
    f("nav start offsets (us)", n0 - base, (1, 50, 90, 100))
    f("nav got state (us)", ns - base)
    f("nav search + tail duration from state (us)", n1 - ns)
    f("nav finish (wait for the word + reward) (us)", n2 - n1)
    f("nav end offsets (us)", n2 - base)
    if os.environ.get("FIN"):     # the finish-role shape (FIN=1): search waves stamp 12 / 5 / 6, finish waves 10 / 9 / 13 / 7 / 15
        f("search: result out (us)", s[:, 6] - base)
        f("search duration from state (us)", s[:, 6] - ns)
        f("finish wave: has its slot (us)", s[:, 10] - base)
        f("finish wave: state + search records seen (us)", s[:, 9] - base)
        f("finish wave: tail duration (us)", n1 - s[:, 9])
        f("finish wave: tail done (us)", n1 - base)
        f("finish wave: wait for the words (us)", s[:, 7] - n1)
        f("finish wave: reward phase + restores (us)", n2 - s[:, 7])
bases = [st[lo:lo + cnt, 3].min() for lo, cnt in env._slices]
print("chain phase offsets of the last step (us):", [round((b - min(bases)) / 100.0, 1) for b in bases])
