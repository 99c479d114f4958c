#!/usr/bin/env python3
"""Are reductions inside a captured CUDA graph replayed correctly on this stack (torch 2.10 + ROCm 7.2)?
Round 3's captured PPO update "read its own gradient norm as inf"; round 4's device-side record of that update shows
|grad| = inf with not one non-finite gradient element, and counts of non-finite inputs (an int64 sum of a bool tensor)
coming back as 4.57e18 = 0x3F83....00000000, i.e. a correct low word under the bits of a float.  This script captures
the reductions of that update step on tensors of its shapes -- nothing of the environment library is involved -- replays
them while other work churns the caching allocator, and compares with eager results."""
import torch

dev = "cuda:0"
torch.manual_seed(0)
N, D = 4096, 186
xs = [torch.randn(N, D, device=dev), torch.randn(N, 2, device=dev), torch.randn(N, device=dev), torch.randn(N, device=dev), torch.randn(N, device=dev)]
gs = [torch.randn(256, 186, device=dev) * 1e-3, torch.randn(256, device=dev) * 1e-3, torch.randn(128, 256, device=dev) * 1e-3,
      torch.randn(128, device=dev) * 1e-3, torch.randn(64, 128, device=dev) * 1e-3, torch.randn(64, device=dev), torch.randn(2, 64, device=dev),
      torch.randn(2, device=dev), torch.randn(2, device=dev)]
out = torch.zeros(4, device=dev, dtype=torch.float64)


def step():
    bad_in = sum((~torch.isfinite(x)).sum() for x in xs)              # int64
    sq = None
    for g in gs:
        s = (g * g).sum()
        sq = s if sq is None else sq + s
    total = torch.sqrt(sq)
    stacked = torch.sqrt(torch.stack([(g * g).sum() for g in gs]).sum())
    out[0].copy_(bad_in), out[1].copy_(total), out[2].copy_(stacked), out[3].copy_(xs[0].abs().max())
    return bad_in


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
ref = out.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
bad = 0
junk = []
for it in range(600):
    # allocator churn of the kind a training loop produces between replays
    junk = [torch.randn(2048, 64 + (it * 7 + k) % 190, device=dev) for k in range(6)]
    _ = torch.cat(junk[:4], 1).sum()
    for x in xs:
        x.normal_()
    for gg in gs:
        gg.normal_().mul_(1e-3)
    g.replay()
    torch.cuda.synchronize()
    want_total = float(torch.sqrt(sum((gg.double() * gg.double()).sum() for gg in gs)))
    got = out.tolist()
    ok = got[0] == 0.0 and abs(got[1] - want_total) < 1e-4 * want_total and abs(got[2] - want_total) < 1e-4 * want_total and \
        abs(got[3] - float(xs[0].abs().max())) < 1e-6
    if not ok:
        bad += 1
        if bad <= 5:
            print("replay %d: bad_in %r total %r stacked %r max %r (expected 0, %r, %r, %r)" % (it, got[0], got[1], got[2], got[3], want_total, want_total, float(xs[0].abs().max())), flush=True)
print("replays with a wrong reduction: %d of 600" % bad)
