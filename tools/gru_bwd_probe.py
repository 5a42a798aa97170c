"""Dev probe: GRU stack backward (step launches and persistent) against nn.GRU in fp64, per gradient."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels, ops
DEV = "cuda:0"
def rel_err(got, ref64):
    got = got.detach().cpu().double(); ref64 = ref64.detach().cpu().double()
    return ((got - ref64).abs().max() / max(1.0, ref64.abs().max().item())).item()
for dims in ((5, 7, 250, 240, 3), (32, 9, 50, 50, 3), (64, 20, 250, 240, 3)):
    B, T, I, H, L = dims
    rnn = torch.nn.GRU(I, H, L, batch_first=True).double()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for p in rnn.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.1 if p.dim() == 1 else 1.0 / math.sqrt(p.shape[1])))
    x = torch.randn(B, T, I, generator=torch.Generator().manual_seed(1))
    x64 = x.double().requires_grad_(True)
    ref, _ = rnn(x64)
    names = [n for l in range(L) for n in ("weight_ih_l%d" % l, "weight_hh_l%d" % l, "bias_ih_l%d" % l, "bias_hh_l%d" % l)]
    dout = torch.randn(B, T, H, generator=torch.Generator().manual_seed(6))
    gref = torch.autograd.grad(ref, [x64] + [getattr(rnn, n) for n in names], dout.double())
    for pers in ("1", "0"):
        os.environ["M2D_GRU_BWD_PERSIST"] = pers
        params = [getattr(rnn, n).detach().float().to(DEV).requires_grad_(True) for n in names]
        xd = x.to(DEV).requires_grad_(True)
        out = ops.gru_stack(xd, params)
        got = torch.autograd.grad(out, [xd] + params, dout.to(DEV))
        print(dims, "persistent bwd" if pers == "1" else "step bwd", "fwd %.1e" % rel_err(out, ref),
              " ".join("%s %.1e" % (n[-9:], rel_err(a, b)) for a, b, n in zip(got, gref, ["x"] + names)), flush=True)
