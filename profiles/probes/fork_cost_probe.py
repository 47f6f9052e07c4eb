#!/usr/bin/env python3
"""What one fork (hipEventRecord on the compute stream + hipStreamWaitEvent on a side stream + a kernel there) costs the
compute stream: a chain of N kernels of ~20 / ~50 us with and without a fork after each (unprofiled, wall time per link)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device('cuda:0')
side = torch.cuda.Stream()
for rows in (20000, 60000, 200000):
    x = torch.randn(rows, 512, device=dev); y = torch.empty_like(x)
    z = torch.randn(4096, 64, device=dev); w = torch.empty_like(z)
    evs = [torch.cuda.Event() for _ in range(256)]
    def chain(mode, n=200):
        for i in range(n):
            torch.mul(x, 1.5, out=y)
            if mode >= 1:
                evs[i % 256].record()
            if mode >= 2:
                side.wait_event(evs[i % 256])
            if mode >= 3:
                with torch.cuda.stream(side):
                    torch.mul(z, 2.0, out=w)
        if mode >= 2:
            torch.cuda.current_stream().wait_stream(side)
    def t(mode):
        chain(mode); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a = time.perf_counter(); chain(mode); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - a) / 200 * 1e6)
        return best
    print('rows %6d: plain %.2f us/link | +record %.2f | +record+wait %.2f | +record+wait+side kernel %.2f' % (rows, t(0), t(1), t(2), t(3)), flush=True)
