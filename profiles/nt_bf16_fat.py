"""Stand-alone timing of the bf16-storage NT products at the fat shapes of the 5-level hierarchy (BASELINE config 5): the
register-staged tiles (STIN_NT_GLDS=0) against the LDS-DMA 128 x 128 kernel (STIN_NT_GLDS=1), interleaved rounds in one process.
    python profiles/nt_bf16_fat.py [--md out.md]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402
dev = torch.device('cuda:0')
SHAPES = [(8100, 4096, 1024), (8100, 1024, 2048), (8100, 2048, 1024), (8100, 1024, 4096), (27000, 2048, 512), (27000, 2560, 512), (27000, 512, 1024),
          (27000, 1024, 512), (27000, 512, 2560), (90000, 1024, 256), (90000, 1280, 256), (90000, 256, 512), (90000, 512, 256), (90000, 256, 1280),
          (300000, 512, 128), (300000, 128, 256)]


def t(f, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    ap.add_argument('--rounds', type=int, default=5)
    args = ap.parse_args()
    lines = ['# bf16-storage NT products, fat shapes (random data, median of %d interleaved rounds x 10 launches; bf16 weights, bias, bf16 out)' % args.rounds, '',
             '| M | Nc | K | register-staged us (TFLOP/s) | LDS-DMA 128x128 | LDS-DMA 256x256 | shipped rule | shipped: frac of 2.5 PF |', '|---|---|---|---|---|---|---|---|']
    for (M, Nc, K) in SHAPES:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(Nc, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(Nc, device=dev)
        out = torch.empty(M, Nc, device=dev, dtype=torch.bfloat16)
        arms = [{'STIN_NT_GLDS': '0'}, {'STIN_NT_GLDS': '1', 'STIN_NT_BIG': '0'}, {'STIN_NT_GLDS': '1', 'STIN_NT_BIG': '1'}, {}]
        res = [[] for _ in arms]
        for r in range(args.rounds + 1):
            for v, env in enumerate(arms):
                for k in ('STIN_NT_GLDS', 'STIN_NT_BIG'):
                    os.environ.pop(k, None)
                os.environ.update(env)
                us = t(lambda: SF.gemm_nt(A, W, b, out=out))
                if r:
                    res[v].append(us)
        for k in ('STIN_NT_GLDS', 'STIN_NT_BIG'):
            os.environ.pop(k, None)
        m = [statistics.median(x) for x in res]
        fl = 2.0 * M * Nc * K
        lines.append('| %d | %d | %d | %s | %.2f |' % (M, Nc, K, ' | '.join('%.1f (%.0f)' % (x, fl / x / 1e6) for x in m), fl / m[3] / 1e6 / 2500))
        print(lines[-1], flush=True)
    if args.md:
        os.makedirs(os.path.dirname(os.path.abspath(args.md)), exist_ok=True)
        open(args.md, 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
