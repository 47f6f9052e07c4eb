"""Stand-alone timing of the bf16-storage TN (weight-gradient) products at the shapes of the 5-level hierarchy: register-transpose
kernel (STIN_TN_TR=0) vs the transposed-read kernel (STIN_TN_TR=1), interleaved rounds.   python profiles/tn_bf16_fat.py [--md out.md]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402
dev = torch.device('cuda:0')
SHAPES = [(8100, 4096, 1024), (8100, 1024, 2048), (27000, 2048, 512), (27000, 2560, 512), (27000, 512, 1024), (90000, 1024, 256), (90000, 1280, 256),
          (90000, 256, 512), (300000, 512, 128), (300000, 128, 256), (18063, 1024, 256), (18063, 256, 512), (161362, 320, 128)]


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
    lines = ['# bf16-storage TN products dW[Nc, K + 1] = G^T [X | 1] incl. the slab reduction (random data, median of %d interleaved rounds x 10 calls)' % args.rounds, '',
             '| M | Nc | K | register transpose, 128 x 128 us (TFLOP/s) | transposed LDS reads, 128 x 128 | shipped rule (256 x 256 tiles where Nc, K >= 512) |', '|---|---|---|---|---|---|']
    for (M, Nc, K) in SHAPES:
        G = torch.randn(M, Nc, device=dev).bfloat16()
        X = torch.randn(M, K, device=dev).bfloat16()
        arms = [{'STIN_TN_TR': '0', 'STIN_TN_BIG': '0'}, {'STIN_TN_TR': '1', 'STIN_TN_BIG': '0'}, {}]
        res = [[], [], []]
        for r in range(args.rounds + 1):
            for v, env in enumerate(arms):
                for k in ('STIN_TN_TR', 'STIN_TN_BIG'):
                    os.environ.pop(k, None)
                os.environ.update(env)
                us = t(lambda: SF.gemm_tn(G, X, ones_column=True))
                if r:
                    res[v].append(us)
        m = [statistics.median(x) for x in res]
        fl = 2.0 * M * Nc * K
        lines.append('| %d | %d | %d | %.1f (%.0f) | %.1f (%.0f) | %.1f (%.0f) |' % (M, Nc, K, m[0], fl / m[0] / 1e6, m[1], fl / m[1] / 1e6,
                                                                           m[2], fl / m[2] / 1e6))
        print(lines[-1], flush=True)
    if args.md:
        os.makedirs(os.path.dirname(os.path.abspath(args.md)), exist_ok=True)
        open(args.md, 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
