"""Diagnostic: per-phase cycle shares of K2/K3 (library must be built with `make STAMPS=1`).
Usage on the GPU box: python bench.py --bank-cache /tmp/bank --steps 20 --cpu-baseline 0; python tools/phase_stamps.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config
from gym_auv_amd.world import build_bank_parallel
cfg = effective_reference_config(use_lidar=True)
n = 4096
z = np.load("/tmp/bank.polygons50.0.4096.4096.2.npz"); bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
env = BatchedAuvEnv(cfg, bank, n, auto_reset=True)
env.reset()
a = torch.rand((n, 2), device="cuda:0") * 2 - 1
for i in range(int(os.environ.get("STEPS", "30"))): env.step(torch.rand((n, 2), device="cuda:0") * torch.tensor([2.0, 0.3], device="cuda:0") - torch.tensor([1.0, 0.15], device="cuda:0"))
torch.cuda.synchronize()
st = env.read("STAMPS").cpu().numpy().astype(np.float64)
names = ["K2.front", "K2.pairs", "K2.back", "-", "-", "K2 active segs", "K2 work items", "K2.pairs stage",
         "K3.bounds", "K3.list", "K3.scan", "K3.nav", "-", "-", "K2.pairs prefix", "K2.pairs items"]
for i, nm in enumerate(names):
    if nm != "-": print("%-18s mean %9.0f  p50 %9.0f  max %9.0f ticks" % (nm, st[:, i].mean(), np.median(st[:, i]), st[:, i].max()) + '  argmax env %d' % st[:, i].argmax())
print(env.step_timed(a))

# wave timeline of K2 (100 MHz wall clock): start/end relative to the first wave start
t0, t1 = st[:, 3], st[:, 4]
base = t0.min()
print("K2 wave start offsets (us): p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile((t0 - base) / 100.0, [50, 90, 100])))
print("K2 wave end offsets   (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile((t1 - base) / 100.0, [50, 90, 99, 100])))
print("K2 wave durations     (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile((t1 - t0) / 100.0, [50, 90, 99, 100])))
late = np.argsort(t1)[-5:]
print("last finishers: env", late, "dur us", (t1[late] - t0[late]) / 100.0, "start us", (t0[late] - base) / 100.0, "segs", st[late, 5], "items", st[late, 6], "stage/prefix/items ticks", st[late, 7], st[late, 14], st[late, 15], "front/pairs/back ticks", st[late, 0], st[late, 1], st[late, 2])

n0, n1 = st[:, 12], st[:, 13]
if n1.max() > 0:
    print("nav wave start offsets (us): p1 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile((n0 - base) / 100.0, [1, 50, 90, 100])))
    print("nav wave end offsets   (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile((n1 - base) / 100.0, [50, 90, 99, 100])))
    print("nav wave durations     (us): p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile((n1 - n0) / 100.0, [50, 90, 100])))
# block retire times of the K2 role (a workgroup frees its slot when its slowest wave ends)
blk = t1.reshape(-1, 4).max(axis=1)
print("K2 block end offsets (us): p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile((blk - base) / 100.0, [10, 50, 90, 100])))
