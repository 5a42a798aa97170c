import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_full_size as F
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
loss, full = F._critic_loss_grads(critic, x_real, x_fake, a, alpha)
loss_b, full_b = F._critic_loss_grads(critic, x_real, x_fake, a, alpha)
l1, g1 = F._critic_loss_grads(critic, x_real[:h].contiguous(), x_fake[:h].contiguous(), a[:h].contiguous(), alpha[:h])
l2, g2 = F._critic_loss_grads(critic, x_real[h:].contiguous(), x_fake[h:].contiguous(), a[h:].contiguous(), alpha[h:])
print("loss", loss.item(), l1.item(), l2.item())
for (n, p), f, fb, p1, p2 in zip(critic.named_parameters(), full, full_b, g1, g2):
    m = 0.5 * (p1 + p2)
    print("%-28s norm %.4e  rel(full,shards) %.2e  rel(full,rerun) %.2e" % (n, f.norm().item(), F.rel(f, m), F.rel(f, fb)))
