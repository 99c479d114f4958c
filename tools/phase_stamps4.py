"""Diagnostic (STAMPS build): wave timelines of the one-launch step (k_step_roles).
Usage on the GPU box:  AUV_HIP_LIB=gym_auv_amd/csrc_stamps/libauv_hip.so python tools/phase_stamps4.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config
cfg = effective_reference_config(use_lidar=True)
n = 4096
z = np.load(os.environ.get("BANK", "/tmp/bank.polygons50.0.4096.4096.2.npz")); bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
env = BatchedAuvEnv(cfg, bank, n, auto_reset=True)
env.set_step_mode("one_launch")
env.reset()
a = torch.rand((n, 2), device="cuda:0") * 2 - 1
for i in range(int(os.environ.get("STEPS", "30"))): env.step(torch.rand((n, 2), device="cuda:0") * torch.tensor([2.0, 0.3], device="cuda:0") - torch.tensor([1.0, 0.15], device="cuda:0"))
torch.cuda.synchronize()
st = env.read("STAMPS").cpu().numpy().astype(np.float64)
pc = lambda x, q: tuple(np.percentile(x, q))
print("timed step (ms):", env.step_timed(a))
k1, l0, lp, ls, l1, l2 = st[:, 0], st[:, 3], st[:, 1], st[:, 2], st[:, 4], st[:, 14]
n0, ns, n1, n2, listed = st[:, 12], st[:, 5], st[:, 13], st[:, 15], st[:, 6]
base = l0.min()
f = lambda nm, x, q=(50, 90, 99, 100): print("%-44s" % nm, " ".join("p%d %.2f" % (qq, v) for qq, v in zip(q, pc(x / 100, list(q)))))
f("dynamics: state published (us)", k1 - base)
f("LiDAR start offsets (us)", l0 - base)
f("LiDAR pre-work done (us)", lp - base)
f("LiDAR state arrived (us)", ls - base)
f("LiDAR sweep end (us)", l1 - base)
f("LiDAR publish after sweep end (us)", l2 - l1)
f("nav start offsets (us)", n0 - base, (1, 50, 90, 100))
f("nav wait for state (us)", ns - n0)
f("nav durations from state (us)", n1 - ns)
f("nav finish (reward) (us)", n2 - n1)
f("nav end offsets (us)", n2 - base)
late = np.arange(n) >= n - n // 8
f("LiDAR sweep end, first 7/8 of the envs (us)", (l1 - base)[~late])
f("LiDAR sweep end, last 1/8 (displaced by the dynamics waves) (us)", (l1 - base)[late])
f("nav end, first 7/8 (us)", (n2 - base)[~late])
f("nav end, last 1/8 (us)", (n2 - base)[late])
f("sweep duration from state, first 7/8 (us)", (l1 - ls)[~late])
f("sweep duration from state, last 1/8 (us)", (l1 - ls)[late])
