"""Per-step duration of the first 60 steps after a reset of the headline workload (all environments in step with each
other: every 25th vessel step ALL of them refresh their nearby-obstacle mask, vessel.py:266-273)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config
cfg = effective_reference_config(use_lidar=True)
n = 4096
z = np.load("/tmp/bank.polygons50.0.4096.4096.2.npz"); bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
env = BatchedAuvEnv(cfg, bank, n, auto_reset=True)
env.reset()
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
ms = []
for i in range(60):
    a = torch.rand((n, 2), device="cuda:0", generator=g) * torch.tensor([2.0, 0.3], device="cuda:0") - torch.tensor([1.0, 0.15], device="cuda:0")
    ms.append(env.step_timed(a)[3] * 1e3)
print(" ".join("%d:%.1f" % (i + 1, m) for i, m in enumerate(ms)))
print("mean of steps 6..25 (the driver's window): %.2f us; steps 25 / 50: %.1f / %.1f us; median of the others: %.2f us" % (np.mean(ms[5:25]), ms[24], ms[49], np.median([m for i, m in enumerate(ms) if (i + 1) % 25])))
