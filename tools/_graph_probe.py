import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MODE = os.environ.get("MODE")
if MODE is None:
    for m in ("torch_only", "conv_fwd", "convbn_fwd", "gemm", "bwd_singlethread", "bwd"):
        r = subprocess.run([sys.executable, __file__], env=dict(os.environ, MODE=m), capture_output=True, text=True, timeout=100)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if "RESULT" in l or "Error" in l or "error" in l][-3:]
        print(m, "rc", r.returncode, tail)
    sys.exit(0)
import torch
from music2dance_amd import ops, kernels
from music2dance_amd.layers import Conv1d, BatchNorm1d
dev = torch.device("cuda:0")
torch.manual_seed(0)
conv = Conv1d(32, 64, 5, stride=2, padding=2).to(dev)
bn = BatchNorm1d(64).to(dev)
x_static = torch.randn(8, 32, 100, device=dev)
K = kernels.impl()
if MODE == "bwd_singlethread":
    torch.autograd.set_multithreading_enabled(False)
def step():
    if MODE == "torch_only":
        return (x_static * 2).sum()
    if MODE == "conv_fwd":
        with torch.no_grad(): return conv(x_static).sum()
    if MODE == "convbn_fwd":
        with torch.no_grad(): return bn(conv(x_static)).sum()
    if MODE == "gemm":
        with torch.no_grad(): return K.gemm(0, x_static.view(8 * 32, 100), x_static.view(8 * 32, 100)).sum()
    for p in list(conv.parameters()) + list(bn.parameters()): p.grad = None
    y = bn(conv(x_static))
    loss = (y * y).mean()
    loss.backward()
    return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("RESULT", MODE, float(out))
