"""Debug aid: every torch.empty device buffer gets a 4 KB guard tail filled with a byte pattern; after a full training step
(whole-block C calls and the per-kernel path) every guard must be intact.  A damaged guard = a kernel wrote past the end of
its output (harmless under the caching allocator's rounding most of the time, a GPU memory fault when the neighbour page is
not writable - seen with HIP-graph replays).

    python profiles/probes/guard.py [--vertices 20000] [--dtype f32|bf16]
"""
import argparse
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
GUARD = 4096
_empty = torch.empty
REG = []


def empty(*size, **kw):
    dev = kw.get('device')
    if dev is None or torch.device(dev).type != 'cuda' or kw.get('pin_memory'):
        return _empty(*size, **kw)
    if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
        size = tuple(size[0])
    dtype = kw.get('dtype') or torch.get_default_dtype()
    n = 1
    for v in size:
        n *= int(v)
    es = torch.empty((), dtype=dtype).element_size()
    # the payload is placed so that it ENDS exactly at the guard (its start stays 256-byte aligned when n*es is a multiple of 256;
    # otherwise pad the front)
    nbytes = n * es
    front = (-nbytes) % 256
    raw = _empty(front + nbytes + GUARD, dtype=torch.uint8, device=dev)
    raw[front + nbytes:].fill_(0xA5)
    t = raw[front:front + nbytes].view(dtype).view(size)
    REG.append((raw, front + nbytes, ''.join(traceback.format_stack(limit=6)[:-1])))
    return t


torch.empty = empty


def check(tag):
    torch.cuda.synchronize()
    bad = 0
    for raw, off, where in REG:
        g = raw[off:]
        if not bool((g == 0xA5).all()):
            nz = (g != 0xA5).nonzero().view(-1)
            print('GUARD DAMAGED (%s): %d bytes, first at +%d, last at +%d of a %d-byte buffer, allocated at:\n%s'
                  % (tag, nz.numel(), int(nz[0]), int(nz[-1]), off, where), flush=True)
            bad += 1
    print('%s: %d buffers checked, %d damaged' % (tag, len(REG), bad), flush=True)
    REG.clear()
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vertices', type=int, default=20000)
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--crops', type=int, default=0)
    args = ap.parse_args()
    from surface_texture_inpainting_net_amd import functional as SF
    from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
    from surface_texture_inpainting_net_amd.data import collate
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9, n_levels=2,
               pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True)
    torch.manual_seed(0)
    net = S.define_G(**cfg).to('cuda:0')
    if args.dtype == 'bf16':
        net.set_activation_dtype(torch.bfloat16)
    step = TrainStep(net)
    if args.crops:
        s = collate([make_synthetic_mesh(args.vertices + 1000 * i, 3, seed=i) for i in range(args.crops)]).to('cuda:0')
    else:
        s = make_synthetic_mesh(args.vertices, 3, seed=0).to('cuda:0')
    REG.clear()
    total = 0
    step(s)
    total += check('whole-block path, step 1')
    s._plan_cache = None
    step(s)
    total += check('whole-block path, step 2 (fresh plan)')
    SF.KernelTimer.start(['none'], max_records=10)
    s._plan_cache = None
    step(s)
    SF.KernelTimer.stop()
    total += check('per-kernel path')
    sys.exit(1 if total else 0)


if __name__ == '__main__':
    main()
