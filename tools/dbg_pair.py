import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
sys.path.insert(0, "tests")
from test_gpu_parity import _mixed_bank, _env
from gym_auv_amd.config import effective_reference_config
n = 64
bank = _mixed_bank(32)
cfg = effective_reference_config(use_lidar=True)
cfg.episode.max_timesteps = 9
ref = _env(cfg, bank, n); ref.reset()
par = _env(cfg, bank, n); par.set_step_mode("paired"); par.reset()
rs = np.random.RandomState(5)
for k in range(3):
    a = torch.as_tensor(rs.uniform([-1, -0.15], [1, 0.15], (n, 2)), dtype=torch.float32, device="cuda:0")
    o0, r0, d0, _ = ref.step(a); o1, r1, d1, _ = par.step(a)
    torch.cuda.synchronize()
    bad = (~(r0 == r1)).nonzero().flatten().cpu().numpy()
    print("step", k, "bad envs", len(bad), bad[:10])
    for f in ("REWARD64", "INFO64", "NAV64", "COUNTERS", "COLLISION", "STEP_INFO"):
        x, y = ref.read(f).cpu().numpy(), par.read(f).cpu().numpy()
        if not np.array_equal(x, y, equal_nan=False):
            i = bad[0] if len(bad) else 0
            print(" ", f, "differs; env", i, "ref", x[i] if x.ndim > 1 and x.shape[0] == n else x[..., i], "par", y[i] if y.ndim > 1 and y.shape[0] == n else y[..., i])
    if len(bad): break
