import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_full_size as F
import music2dance_amd.losses as L
from music2dance_amd import ops
import bench
from music2dance_amd.engine import synthetic_phase3_batch
B, T, DEV = 64, 120, "cuda:0"
gen, critic = bench.build_models(torch.device(DEV), T)
real, audio, _ = synthetic_phase3_batch(B, T, torch.device(DEV), seed=5)
g = torch.Generator().manual_seed(9)
x_real = real.permute(0, 2, 1).contiguous()
x_fake = torch.rand(B, 69, T, generator=g).to(DEV)
a = audio.unsqueeze(1)
alpha = torch.rand(B, 1, generator=g)
h = B // 2
def grads(xr, xf, au, al, mode):
    n = xr.size(0)
    orig = torch.rand
    torch.rand = lambda *aa, **k: al.cpu().clone()
    try:
        critic.zero_grad(set_to_none=True)
        au = au.clone()
        with critic.shared_audio():
            loss = 0
            if mode in ("gp", "all"):
                loss = loss + 10.0 * L.gradient_penalty(critic, n, xr, xf, au, is_seq=True, lp=False, device=xr.device)
            if mode in ("w", "all"):
                s_real, s_fake = critic.score_pair(xr, xf, au)
                loss = loss + s_fake.mean() - s_real.mean()
            loss.backward()
    finally:
        torch.rand = orig
    return [None if p.grad is None else p.grad.detach().clone() for p in critic.parameters()]
for mode in ("w", "gp"):
    full = grads(x_real, x_fake, a, alpha, mode)
    g1 = grads(x_real[:h].contiguous(), x_fake[:h].contiguous(), a[:h].contiguous(), alpha[:h], mode)
    g2 = grads(x_real[h:].contiguous(), x_fake[h:].contiguous(), a[h:].contiguous(), alpha[h:], mode)
    print("==== mode", mode)
    for (n, p), f, p1, p2 in zip(critic.named_parameters(), full, g1, g2):
        if f is None: print(n, None); continue
        print("%-30s max %.3e rel %.2e" % (n, f.abs().max().item(), F.rel(f, 0.5 * (p1 + p2))))
# per-sample gradient penalty norms: full vs shards
def norms(xr, xf, au, al):
    n = xr.size(0)
    interp = (al.to(DEV) * xr.reshape(n, -1) + (1 - al.to(DEV)) * xf.reshape(n, -1)).view(n, 69, -1).requires_grad_(True)
    au = au.clone().requires_grad_(True)
    s = critic(interp, au)
    g0, g1 = torch.autograd.grad(s, (interp, au), torch.ones_like(s))
    return g0.reshape(n, -1).norm(dim=1), g1.reshape(n, -1).norm(dim=1)
nf = norms(x_real, x_fake, a, alpha)
n1 = norms(x_real[:h].contiguous(), x_fake[:h].contiguous(), a[:h].contiguous(), alpha[:h])
print("pose-grad norms rel", F.rel(nf[0][:h], n1[0]), "audio-grad norms rel", F.rel(nf[1][:h], n1[1]))
print(nf[0][:4], nf[1][:4])
