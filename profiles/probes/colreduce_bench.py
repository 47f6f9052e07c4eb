"""Stand-alone timing of the instance-norm reductions at the headline shapes (DOT_ELU: reads agg + gout; MOMENTS: reads agg)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd.plan import NormGroups
dev = torch.device('cuda:0')
def t(f, n=30):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for N, C in ((200704, 64), (60211, 128), (18063, 256)):
    x, g = torch.randn(N, C, device=dev), torch.randn(N, C, device=dev)
    ng = NormGroups(N, dev)
    mean, rstd = SF.instance_stats(x, ng)
    us_m = t(lambda: SF.colreduce(SF.RED_MOMENTS, x, ng, ng.ptr_sum))
    us_d = t(lambda: SF.colreduce(SF.RED_DOT_ELU, x, ng, ng.ptr_true, gout=g, mean=mean, rstd=rstd, post=SF.POST_NORM_COEF))
    mb = N * C * 4 / 1e6
    print('N=%d C=%d: MOMENTS %.1f us (%.0f MB -> %.2f TB/s), DOT_ELU %.1f us (%.0f MB -> %.2f TB/s)  [each incl. its final kernel]'
          % (N, C, us_m, mb, mb / us_m, us_d, 2 * mb, 2 * mb / us_d))
