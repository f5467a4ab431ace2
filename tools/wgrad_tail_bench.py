"""Kernel time of the layer tail's weight-gradient group (dense 64x64, feed-forward 64->256 and 256->64) and of its
members alone (run under rocprofv3 --kernel-trace; fold with tools/kstats.py or the trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ac_tsr_amd import ops
M = 25600
r = lambda *s: torch.randn(*s, device="cuda")
dense = (r(M, 64), r(M, 64), True)
ff1 = (r(M, 64), r(M, 256), True)
ff2 = (r(M, 256), r(M, 64), True)
for group in ([dense, ff1, ff2], [dense], [ff1], [ff2], [ff1, ff2]):
    for _ in range(20):
        ops.linear_wgrad_grouped(group)
    torch.cuda.synchronize()
