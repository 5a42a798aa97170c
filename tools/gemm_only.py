"""Dev tool: the engine on plain 4096-cubed GEMMs only (NT / NN / TN), for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
n = 4096
a = torch.randn(n, n, device="cuda"); b = torch.randn(n, n, device="cuda")
for _ in range(3):
    K.gemm(1, a, b)
x = torch.randn(64, 64, 4800, device="cuda"); w = torch.randn(128, 64, 25, device="cuda") * 0.02
for _ in range(3):
    K.conv1d_fwd(x, w, None, 4, 11)
torch.cuda.synchronize()
