for kc in 128 64; do echo "== KC=$kc"; STIN_NT_STREAM_KC=$kc python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from surface_texture_inpainting_net_amd import functional as SF
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for m, nc, k in [(1200642, 64, 128), (361000, 128, 256), (361000, 256, 128), (200704, 64, 128), (200704, 256, 128)]:
    g = torch.Generator().manual_seed(1)
    A = (torch.randn(m, k, generator=g) * 0.7).to('cuda:0'); W = (torch.randn(nc, k, generator=g) * 0.1).to('cuda:0')
    mean = torch.zeros(k, device='cuda:0'); one = torch.ones(k, device='cuda:0')
    out = torch.empty(m, nc, device='cuda:0'); st = SF._stream(A)
    r = []
    for nt in ('2', '4'):
        os.environ['STIN_NT_STREAM_NT'] = nt
        t0 = timeit(lambda: SF._call('stin_gemm_nt_stream_f32', SF._ptr(A), k, SF._ptr(W), k, None, None, None, None, m, nc, k, SF._ptr(out), nc, int(SF.PREC_FWD), st))
        t1 = timeit(lambda: SF._call('stin_gemm_nt_stream_f32', SF._ptr(A), k, SF._ptr(W), k, SF._ptr(mean), SF._ptr(one), SF._ptr(one), SF._ptr(mean), m, nc, k, SF._ptr(out), nc, int(SF.PREC_FWD), st))
        r.append('NT=%s plain %.0f bn %.0f' % (nt, t0, t1))
    print(m, nc, k, ' | '.join(r), flush=True)
PY
done
