"""Dev tool: is the hand-scheduled critic iteration as accurate as the autograd path? Both HIP paths against an fp64
evaluation of the oracle on the host, per parameter tensor: max |g - g64| / max |g64| (BASELINE configs[2] critic, B = 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from music2dance_amd import kernels
from music2dance_amd.critic_step import CriticStep
from oracle import m2d_oracle as O
from tests.test_critic_step import _autograd, _inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
_, critic = bench.build_models(dev, 120)
real, fake_rows, alpha, audio = _inputs(B, 120, dev, seed=9)
torch.set_num_threads(min(64, os.cpu_count() or 8))
dsd = {k: v.detach().cpu().double() for k, v in critic.state_dict().items()}
d_params, _ = O.split_state(dsd)
real_c = real.cpu().double().permute(0, 2, 1).contiguous()
fake_c = fake_rows.cpu().double().view(B, 120, 69).permute(0, 2, 1).contiguous()
aud = audio.cpu().double()
crit = lambda x, a=None: O.p3_critic(d_params, x, a, 25, "id", False)
gp, _, _ = O.gradient_penalty(crit, real_c, fake_c, alpha.cpu().double(), aud.clone(), is_seq=True, lp=False)
loss = crit(fake_c, aud).mean() - crit(real_c, aud).mean() + 10.0 * gp
g64 = O.grads_of(loss, d_params)
with kernels.impl().weight_cache():
    _, g_auto = _autograd(critic, real, fake_rows, alpha, audio, 10.0, False)
    CriticStep(critic, 10.0).run(real, fake_rows, audio.clone(), alpha)
torch.cuda.synchronize()
print("%-34s %10s %12s %12s" % ("tensor", "max|g64|", "autograd", "manual"))
for n, p in critic.named_parameters():
    ref = g64[n]
    s = ref.abs().max().item()
    ea = (g_auto[n].cpu().double() - ref).abs().max().item() / s
    em = (p.grad.cpu().double() - ref).abs().max().item() / s
    print("%-34s %10.3e %12.2e %12.2e" % (n, s, ea, em))
print("loss fp64 %.6f gp %.6f" % (loss.item(), gp.item()))
