"""Kernel time of acattn_linear_wgrad_grouped against the number of items (run under rocprofv3 by gpu_wgrad_group.sh)."""
import torch
from ac_tsr_amd import ops
M = 25600
xs = [torch.randn(M, 64, device="cuda") for _ in range(8)]
gs = [torch.randn(M, 64, device="cuda") for _ in range(8)]
for n in (1, 2, 4, 6, 8):
    for _ in range(20):
        ops.linear_wgrad_grouped([(xs[i], gs[i], True) for i in range(n)])
    torch.cuda.synchronize()
