import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/examples")
import ppo
dev = "cuda:0"
torch.manual_seed(0)
net = ppo.ActorCritic(186).to(dev)
params = list(net.parameters())
pi_params, v_params = list(net.pi.parameters()) + [net.log_std], list(net.v.parameters())
opt = torch.optim.Adam(params, lr=2e-4, capturable=True)
N = 2048
def step(o, a, lp, adv, ret):
    mu = net.pi(o)
    ratio = (net.log_prob(mu, a) - lp).exp()
    pg = -torch.min(ratio * adv, ratio.clamp(0.8, 1.2) * adv).mean()
    vf = 0.5 * (net.v(o).squeeze(-1) - ret).pow(2).mean()
    loss = pg + 0.5 * vf - 0.001 * net.entropy()
    opt.zero_grad(set_to_none=False)
    loss.backward()
    ppo.clip_grad_norm(pi_params, 0.5); ppo.clip_grad_norm(v_params, 0.5)
    opt.step()
    return loss
def data():
    o = torch.randn(N, 186, device=dev); a = torch.randn(N, 2, device=dev)
    with torch.no_grad():
        lp = net.log_prob(net.pi(o), a)
    return [o, a, lp, torch.randn(N, device=dev), torch.randn(N, device=dev)]
inp = data()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step(*inp)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step(*inp)
w = lambda: float(sum(p.detach().abs().sum() for p in pi_params))
print("after capture", w())
for it in range(12):
    new = data()
    for b, s in zip(inp, new): b.copy_(s)
    g.replay()
    torch.cuda.synchronize()
    print(it, "loss %.5f" % float(loss), "|w| %.6f" % w(), "std", net.log_std.exp().tolist())

# ---- back-to-back replays without host synchronisation, inputs refreshed by index_select(out=) as examples/ppo.py does
big = [torch.cat([data()[k] for _ in range(8)]) for k in range(5)]
for mode in ("sync", "nosync", "nosync"):
    w0 = w()
    for ep in range(4):
        perm = torch.randperm(big[0].shape[0], device=dev)
        for mb in perm.chunk(8):
            for b, s in zip(inp, big):
                torch.index_select(s, 0, mb, out=b)
            g.replay()
            if mode == "sync":
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(mode, "32 replays: |w| %.6f -> %.6f" % (w0, w()))
