"""Debug aid: run the model with every torch.empty / empty_like device buffer pre-filled with a poison pattern
(0xFF bytes = NaN for fp32 / bf16, -1 for integers, or a second pattern 0x3C = finite garbage) and compare outputs and
gradients bit for bit with an unpoisoned run.  Any difference = some kernel consumes memory nobody wrote."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh

DEV = 'cuda:0'
_empty, _empty_like = torch.empty, torch.empty_like
PATTERN = [None]


def _poison(t):
    if PATTERN[0] is not None and t.is_cuda and t.numel() > 0:
        t.view(torch.uint8).fill_(PATTERN[0]) if t.is_contiguous() else None
    return t


def empty(*a, **k):
    return _poison(_empty(*a, **k))


def empty_like(*a, **k):
    return _poison(_empty_like(*a, **k))


torch.empty, torch.empty_like = empty, empty_like


def run(cfg, n, levels, dtype, dil):
    torch.manual_seed(7)
    net = S.define_G(**cfg).to(DEV)
    if dtype == 'bf16':
        net.set_activation_dtype(torch.bfloat16)
    s = make_synthetic_mesh(n, levels, seed=12, dilations=dil).to(DEV)
    params = list(net.parameters())
    names = [k for k, _ in net.named_parameters()]

    def once():
        net.zero_grad(set_to_none=True)
        out = net(s)
        out.float().square().mean().backward()
        return [out.detach().clone()] + [p.grad.clone() for p in params]

    PATTERN[0] = None
    want = once()
    for pat in (0xFF, 0x3C, 0x7F):
        PATTERN[0] = pat
        s._plan_cache = None
        got = once()
        PATTERN[0] = None
        for k, a, b in zip(['out'] + names, got, want):
            if not torch.equal(a, b):
                print('  MISMATCH pattern 0x%02X' % pat, k, tuple(a.shape), 'nan' if a.isnan().any() else float((a.float() - b.float()).abs().max()), flush=True)
    print('checked', cfg.get('filter_type'), cfg.get('norm'), n, dtype, flush=True)


base = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
            pooling_type='max', dilations=[1, 2, 4])
run(base, 40_000, 3, 'f32', (2, 4))
run(dict(base, pooling_type='mean'), 12_345, 3, 'f32', (2, 4))
run(base, 40_000, 3, 'bf16', (2, 4))
run(dict(base, ngf=32), 20_000, 3, 'f32', (2, 4))
run(dict(base, filter_type='edgeconv', norm='batch'), 20_000, 3, 'f32', (2, 4))
run(dict(base, filter_type='sageconv'), 20_000, 3, 'f32', (2, 4))
print('done')
