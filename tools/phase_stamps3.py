"""Diagnostic (STAMPS build): wave timelines of the side-by-side launch, plain or paired (MODE=side_by_side|paired).
Usage on the GPU box:  AUV_HIP_LIB=gym_auv_amd/csrc_stamps/libauv_hip.so MODE=paired python tools/phase_stamps3.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config
cfg = effective_reference_config(use_lidar=True)
n = 4096
z = np.load("/tmp/bank.polygons50.0.4096.4096.2.npz"); bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
env = BatchedAuvEnv(cfg, bank, n, auto_reset=True)
mode = os.environ.get("MODE", "paired")
env.set_step_mode(mode)
env.reset()
a = torch.rand((n, 2), device="cuda:0") * 2 - 1
for i in range(int(os.environ.get("STEPS", "30"))): env.step(torch.rand((n, 2), device="cuda:0") * torch.tensor([2.0, 0.3], device="cuda:0") - torch.tensor([1.0, 0.15], device="cuda:0"))
torch.cuda.synchronize()
st = env.read("STAMPS").cpu().numpy().astype(np.float64)
pc = lambda x, q: tuple(np.percentile(x, q))
print(mode, "timed step (ms):", env.step_timed(a))
t0, t1, n0, n1 = st[:, 3], st[:, 4], st[:, 12], st[:, 13]
base = t0.min()
print("LiDAR start offsets (us): p50 %.1f p90 %.1f max %.1f" % pc((t0 - base) / 100, [50, 90, 100]))
print("LiDAR sweep end     (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % pc((t1 - base) / 100, [50, 90, 99, 100]))
print("nav start offsets   (us): p1 %.1f p50 %.1f p90 %.1f max %.1f" % pc((n0 - base) / 100, [1, 50, 90, 100]))
print("nav durations       (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % pc((n1 - n0) / 100, [50, 90, 99, 100]))
print("nav end offsets     (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % pc((n1 - base) / 100, [50, 90, 99, 100]))
for i, nm in ((0, "K2.front"), (1, "K2.pairs"), (2, "K2.back"), (8, "nav.bounds"), (9, "nav.list"), (10, "nav.scan"), (11, "nav.tail")):
    print("  %-12s cycles: mean %8.0f p50 %8.0f p99 %8.0f max %8.0f" % ((nm, st[:, i].mean()) + pc(st[:, i], [50, 99, 100])))
if mode == "paired":
    le, ne, f0 = st[:, 14], st[:, 15], st[:, 7]
    print("LiDAR publish (after sweep end)           (us): p50 %.2f p90 %.2f p99 %.2f max %.2f" % pc((le - t1) / 100, [50, 90, 99, 100]))
    print("nav: wait for the sweep's word            (us): p50 %.2f p90 %.2f p99 %.2f max %.2f" % pc((f0 - n1) / 100, [50, 90, 99, 100]))
    print("nav: reward phase                         (us): p50 %.2f p90 %.2f p99 %.2f max %.2f" % pc((ne - f0) / 100, [50, 90, 99, 100]))
    print("nav waves that ended before their sweep: %d" % int((n1 < le).sum()))
    print("last end of all waves (us): %.1f" % ((max(le.max(), ne.max()) - base) / 100))
else:
    print("last end of all waves (us): %.1f" % ((max(t1.max(), n1.max()) - base) / 100))
late = np.arange(n) >= n - n // 8
seg = st[:, 5]
print("active boundary segments per env: first 7/8 mean %.1f, last 1/8 mean %.1f" % (seg[~late].mean(), seg[late].mean()))
print("sweep duration (us): first 7/8 p50 %.2f, last 1/8 p50 %.2f" % (np.median((t1 - t0)[~late]) / 100, np.median((t1 - t0)[late]) / 100))
for k in range(8):
    m = (np.arange(n) >= k * n // 8) & (np.arange(n) < (k + 1) * n // 8)
    print("  envs [%4d, %4d): segs %.1f  sweep p50 %.2f us" % (k * n // 8, (k + 1) * n // 8, seg[m].mean(), np.median((t1 - t0)[m]) / 100))
