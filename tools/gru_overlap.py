"""Dev tool: does the persistent GRU forward run UNDER a stream of big conv GEMMs (the generator forward of a critic
iteration runs on its own stream under the critic's kernels), or does it wait for the chip to drain?
Stream A: N forward convs of the audio critic's third layer (~0.33 ms each); stream B: one (64, 120, 240) x 3-layer
GRU forward, enqueued first. Reports A alone, B alone, both together."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
k = kernels.impl()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
B, T, H, L = 64, 120, int(os.environ.get("H", 240)), int(os.environ.get("L", 3))
gi0 = (torch.randn(B, T, 3 * H, generator=g) * 0.5).to(dev)
w_hh_t = [(torch.randn(H, 3 * H, generator=g) / math.sqrt(H)).to(dev) for _ in range(L)]
w_ih_t = [None] + [(torch.randn(H, 3 * H, generator=g) / math.sqrt(H)).to(dev) for _ in range(L - 1)]
b_hh = [(torch.randn(3 * H, generator=g) * 0.1).to(dev) for _ in range(L)]
b_ih = [None] + [(torch.randn(3 * H, generator=g) * 0.1).to(dev) for _ in range(L - 1)]
x = torch.randn(64, 64, 4800, generator=g).to(dev)
w = (torch.randn(128, 64, 25, generator=g) * 0.02).to(dev)
bias = torch.zeros(128, device=dev)
N = 12
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def convs():
    for _ in range(N): k.conv1d_fwd(x, w, bias, 4, 11, 1)
def gru():
    k.gru_stack_fwd(gi0, w_ih_t, b_ih, w_hh_t, b_hh, None, False, persistent=True)
def timed_late(after):
    """the GRU becomes runnable only after `after` convs have gone by on stream A (the chip is full of GEMM workgroups)"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eb, es = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_event(e0)
    with torch.cuda.stream(sa):
        for i in range(N):
            k.conv1d_fwd(x, w, bias, 4, 11, 1)
            if i == after - 1:
                es.record()
                sb.wait_event(es)
                with torch.cuda.stream(sb): gru(); eb.record()
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), e0.elapsed_time(es), e0.elapsed_time(eb)


def timed(fa, fb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eb = torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_event(e0); sb.wait_event(e0)
    if fb:
        with torch.cuda.stream(sb): fb(); eb.record()
    if fa:
        with torch.cuda.stream(sa): fa()
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), (e0.elapsed_time(eb) if fb else 0.0)
with k.weight_cache():
    for _ in range(2): timed(convs, gru)
    for name, fa, fb in (("convs alone", convs, None), ("gru alone", None, gru), ("together", convs, gru), ("together", convs, gru)):
        tot, tb = timed(fa, fb)
        print("%-12s total %.3f ms   gru done at %.3f ms" % (name, tot, tb), flush=True)
    for _ in range(3):
        tot, ts, tb = timed_late(3)
        print("late start   total %.3f ms   gru runnable at %.3f ms, done at %.3f ms" % (tot, ts, tb), flush=True)
k.check_async_errors()
