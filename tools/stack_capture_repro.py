#!/usr/bin/env python3
"""Does a torch.stack / torch.cat captured in a CUDA graph survive other cat / stack calls on this stack (ROCm, torch 2.10)?
ADVICE r3: round 3's captured PPO update "read its own gradient norm as inf".  The norm was sqrt(stack([...]).sum()); if the
stack kernel's table of input addresses is staged through pinned HOST memory by an asynchronous copy, the capture records
that copy with the host address, and a replay uploads whatever the recycled buffer holds by then.  This script captures
y = stack(nine scalars).sum(), replays it while unrelated cat / stack calls churn the host allocator, and compares with
the sum computed without stack inside the same graph."""
import torch

dev = "cuda:0"
torch.manual_seed(0)
xs = [torch.full((), float(i + 1), device=dev) for i in range(9)]        # 1 + 2 + ... + 9 = 45
big = [torch.randn(64, 7, device=dev) for _ in range(40)]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        a = torch.stack(xs).sum()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y_stack = torch.stack(xs).sum()
    acc = xs[0]
    for x in xs[1:]:
        acc = acc + x
    y_plain = acc
bad = 0
for it in range(400):
    # churn: cats / stacks of many inputs, of other sizes, outside the graph
    for k in (3, 9, 17, 40):
        torch.cat(big[:k], 1)
        torch.stack([b.sum() for b in big[:k]])
    g.replay()
    torch.cuda.synchronize()
    a, b = float(y_stack), float(y_plain)
    if a != 45.0 or b != 45.0:
        bad += 1
        if bad <= 5:
            print("replay %d: stack-based %r, plain %r (expected 45.0)" % (it, a, b), flush=True)
print("replays with a wrong stack-based sum: %d of 400; plain sums wrong: see above" % bad)
