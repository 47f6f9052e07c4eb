#!/usr/bin/env python3
"""Experiment: how long do the bottleneck level's weight-gradient products take when each output tile streams ALL rows (no row
split, STIN_TN_BLOCKS small) and nine blocks' products run side by side on nine streams, against the shipped one-block-at-a-time
form?  NB=9 SHAPES="M,Nc,K;..." python profiles/probes/tn_batch_probe.py"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF
shapes = [tuple(int(v) for v in s.split(',')) for s in os.environ.get('SHAPES', '18063,1024,256;18063,256,512').split(';')]
NB = int(os.environ.get('NB', '9'))
streams = [torch.cuda.Stream() for _ in range(NB)]
ops = []
for b in range(NB):
    for M, Nc, K in shapes:
        ops.append((torch.randn(M, Nc, device='cuda'), torch.randn(M, K, device='cuda')))
torch.cuda.synchronize()
def seq():
    for G, X in ops: SF.gemm_tn(G, X, ones_column=True, precision=SF.PREC_BWD)
def par():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    for i, (G, X) in enumerate(ops):
        s = streams[(i // len(shapes)) % NB]
        s.wait_event(ev)
        with torch.cuda.stream(s): SF.gemm_tn(G, X, ones_column=True, precision=SF.PREC_BWD)
    for s in streams: cur.wait_stream(s)
def t(f, n=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        a.record()
        for _ in range(n): f()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return statistics.median(ts), min(ts)
print('STIN_TN_BLOCKS=%s MAXROWS=%s NB=%d shapes=%s' % (os.environ.get('STIN_TN_BLOCKS'), os.environ.get('STIN_TN_MAXROWS'), NB, shapes))
print('  sequential on one stream: median %.1f us min %.1f (all %d products)' % (*t(seq), len(ops)))
print('  side by side on %d streams: median %.1f us min %.1f' % (NB, *t(par)), flush=True)
