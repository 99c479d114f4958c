"""Diagnostic (STAMPS build, two-kernel step): wave timelines of k1n_dyn_nav and k2r_lidar_reward.
Usage on the GPU box:  AUV_HIP_LIB=gym_auv_amd/csrc_stamps/libauv_hip.so python tools/phase_stamps2.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config
cfg = effective_reference_config(use_lidar=True)
n = 4096
z = np.load("/tmp/bank.polygons50.0.4096.4096.2.npz"); bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
env = BatchedAuvEnv(cfg, bank, n, auto_reset=True)
env.set_step_mode(os.environ.get("MODE", "two_kernels"))
env.reset()
a = torch.rand((n, 2), device="cuda:0") * 2 - 1
for i in range(int(os.environ.get("STEPS", "30"))): env.step(torch.rand((n, 2), device="cuda:0") * torch.tensor([2.0, 0.3], device="cuda:0") - torch.tensor([1.0, 0.15], device="cuda:0"))
torch.cuda.synchronize()
st = env.read("STAMPS").cpu().numpy().astype(np.float64)
pc = lambda x, q: tuple(np.percentile(x, q))
print("timed step (ms):", env.step_timed(a))
wg0, n0, n1 = st[:, 14], st[:, 12], st[:, 13]
base = wg0.min()
print("k1n  WG start offsets   (us): p50 %.1f p90 %.1f max %.1f" % pc((wg0 - base) / 100, [50, 90, 100]))
print("k1n  K1+barrier         (us): p50 %.1f p90 %.1f max %.1f" % pc((n0 - wg0) / 100, [50, 90, 100]))
print("k1n  nav durations      (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % pc((n1 - n0) / 100, [50, 90, 99, 100]))
print("k1n  nav end offsets    (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % pc((n1 - base) / 100, [50, 90, 99, 100]))
for i, nm in ((8, "nav.bounds"), (9, "nav.list"), (10, "nav.scan"), (11, "nav.tail")):
    print("  %-12s cycles: mean %8.0f p50 %8.0f p99 %8.0f max %8.0f" % ((nm, st[:, i].mean()) + pc(st[:, i], [50, 99, 100])))
t0, t1, t2 = st[:, 3], st[:, 4], st[:, 15]
b2 = t0.min()
print("k2r  wave start offsets (us): p50 %.1f p90 %.1f max %.1f" % pc((t0 - b2) / 100, [50, 90, 100]))
print("k2r  sweep durations    (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % pc((t1 - t0) / 100, [50, 90, 99, 100]))
print("k2r  reward phase       (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % pc((t2 - t1) / 100, [50, 90, 99, 100]))
print("k2r  wave end offsets   (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % pc((t2 - b2) / 100, [50, 90, 99, 100]))
for i, nm in ((0, "K2.front"), (1, "K2.pairs"), (2, "K2.back")):
    print("  %-12s cycles: mean %8.0f p50 %8.0f p99 %8.0f max %8.0f" % ((nm, st[:, i].mean()) + pc(st[:, i], [50, 99, 100])))
print("gap k1n last end -> k2r first start (us): %.1f" % ((b2 - n1.max()) / 100))
seg = st[:, 5]
print("active boundary segments per env: p50 %.0f p90 %.0f p99 %.0f max %.0f" % pc(seg, [50, 90, 99, 100]))
dur = (t1 - t0) / 100
for lo, hi in ((0, 1), (1, 32), (32, 64), (64, 128), (128, 192), (192, 100000)):
    m = (seg >= lo) & (seg < hi)
    if m.any():
        print("  segs [%d, %d): %5d envs, sweep us p50 %.1f max %.1f | front %.0f pairs %.0f back %.0f cycles (mean)" % (lo, hi, m.sum(), np.median(dur[m]), dur[m].max(), st[m, 0].mean(), st[m, 1].mean(), st[m, 2].mean()))
