"""Micro-costs of the host-side pieces of one fused block call (1000 repetitions each)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF, _lib
from surface_texture_inpainting_net_amd.plan import _ptr, _stream
dev = torch.device('cuda:0')
lib = _lib.load()
def t(name, f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n): f()
    b = time.perf_counter()
    torch.cuda.synchronize()
    print('%-46s %.2f us' % (name, (b - a) / n * 1e6))
x = torch.empty(20000, 256, device=dev)
W = torch.nn.Parameter(torch.empty(512, 256, device=dev))
t('torch.empty([20000,1024]) cuda', lambda: torch.empty(20000, 1024, dtype=torch.float32, device=dev))
t('torch.empty(uint8 ws)', lambda: torch.empty(123456, dtype=torch.uint8, device=dev))
t('_ptr(x)', lambda: _ptr(x))
t('_stream(x)', lambda: _stream(x))
t('W.contiguous()', lambda: W.contiguous())
t('x.stride(0)', lambda: x.stride(0))
t('lib.stin_version()', lambda: lib.stin_version())
t('workspace_bytes ctypes call', lambda: lib.stin_edgeconv_block_fwd_workspace_bytes(256, 256, 512, 256, 0, 1))
t('torch.cuda.current_stream().cuda_stream', lambda: torch.cuda.current_stream(dev).cuda_stream)
t('getattr(lib, name)', lambda: getattr(lib, 'stin_edgeconv_block_fwd'))
a, b = torch.empty(20000, 64, device=dev), torch.empty(20000, 64, device=dev)
t('a.copy_(b) launch', lambda: a.copy_(b), 500)
