"""Dev tool: GRU stack forward and backward, per-step launches vs the persistent launch."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
k = kernels.impl()
DEV = "cuda:0"
def case(B, T, H, L):
    g = torch.Generator().manual_seed(3)
    gi0 = (torch.randn(B, T, 3 * H, generator=g) * 0.5).to(DEV)
    w_hh_t = [(torch.randn(H, 3 * H, generator=g) / math.sqrt(H)).to(DEV) for _ in range(L)]
    w_ih_t = [None] + [(torch.randn(H, 3 * H, generator=g) / math.sqrt(H)).to(DEV) for _ in range(L - 1)]
    b_hh = [(torch.randn(3 * H, generator=g) * 0.1).to(DEV) for _ in range(L)]
    b_ih = [None] + [(torch.randn(3 * H, generator=g) * 0.1).to(DEV) for _ in range(L - 1)]
    for pers in (False, True):
        f = lambda: k.gru_stack_fwd(gi0, w_ih_t, b_ih, w_hh_t, b_hh, None, True, persistent=pers)
        f(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): f()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print("B%d T%d H%d L%d persistent=%s: %.3f ms = %.2f us per step" % (B, T, H, L, pers, ms, 1e3 * ms / (T + L - 1)), flush=True)
    # backward through time (round 3: persistent form)
    outs, saved = k.gru_stack_fwd(gi0, w_ih_t, b_ih, w_hh_t, b_hh, None, True, persistent=False)
    w_hh = [w.t().contiguous() for w in w_hh_t]
    w_ih = [None] + [w.t().contiguous() for w in w_ih_t[1:]]
    dout = torch.randn(B, T, H, generator=g).to(DEV)
    for pers in (False, True):
        f = lambda: k.gru_stack_bwd(dout, outs, saved, w_hh, w_ih, None, persistent=pers)
        f(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): f()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print("  backward persistent=%s: %.3f ms = %.2f us per step" % (pers, ms, 1e3 * ms / (T + L - 1)), flush=True)
    k.check_async_errors()
case(64, 120, 240, 3); case(16, 300, 240, 3); case(64, 120, 10, 1); case(32, 120, 50, 3); case(64, 120, 240, 1)
