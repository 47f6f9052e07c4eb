#!/usr/bin/env python3
"""Headline benchmark: vertices/sec, forward+backward(+optimizer) of SurfaceTextureInpaintingNet on a
synthetic ScanNet-sized mesh (BASELINE.json: 200k vertices / 1.2M directed edges, 3 graph levels, fp32,
the shipped 3-D config), one scene per GPU, pure data parallel with one RCCL gradient all-reduce.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...          # N > 1 without a launcher: starts N ranks itself (torch.distributed.run child,
                                          # before this process has touched the GPU) and exits with their status
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A "step" = GraphPlan (CSR) build from the int64 index tensors + forward + masked weighted L1 loss +
backward + flat-bucket gradient all-reduce + Adam(amsgrad) update, on inputs already resident in HBM.
Rank 0 prints ONE JSON line of < 4 KB (`compact_line`: the contract keys, `roofline`, `cpu_baseline`, a few scalars - no prose) and
writes the full detail object (every secondary leg, per-kernel tables, notes) to `--detail` (default gpurun_out/bench_detail.json);
see DESIGN.md §Measurement for the roofline / cpu_baseline definitions.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import surface_texture_inpainting_net_amd  # noqa: E402
if '--graph' in sys.argv:                       # opt-in ROCm 7.2 graph-replay workaround: must precede the first GPU call
    surface_texture_inpainting_net_amd.enable_graph_replay()

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md, chip-level parameters)
MFMA_16BIT_PEAK_TF = 2500.0    # dense bf16 / f16 MFMA peak (same guide); the fp32-storage GEMMs issue 3 such MFMAs per product
MFMA_F32_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 peak = the rate an exact-fp32 GEMM could reach
HBM_COPY_GBS = 6290.0          # achievable HBM copy rate measured in the guide (used for the GEMMs' byte-side lower bound)

CONFIG_3D = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9,
                 n_levels=2, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True,
                 num_blocks_per_uncheckpointed_block=1)

COMPACT_LIMIT = 4096           # the driver keeps a bounded tail of stdout: the final line must fit in it with room to spare


def _num(v, digits=6):
    """Round floats for the compact line (6 significant digits); pass None / ints / strings through."""
    if isinstance(v, float):
        return float('%.*g' % (digits, v))
    return v


def compact_line(out, detail_path=None):
    """The ONE stdout line: the bench contract's keys + `roofline` + `cpu_baseline` + a few scalars of the secondary legs, all
    numbers or short identifiers (round 5's line carried paragraphs of notes, grew to 20 KB and the driver's 8 KB tail lost its
    head).  `out` is the full detail object, which goes to a file instead."""
    cfg = out.get('config', {})
    line = {k: _num(out.get(k)) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                          'scaling', 'vs_baseline', 'dtype', 'data')}
    line['config'] = {'workload': 'STINet 3-D config, synthetic mesh %s vertices / %s directed edges / %s levels per GPU, step = plan build + '
                                  'fwd + L1 + bwd + all-reduce + Adam' % (cfg.get('vertices_per_gpu'), cfg.get('edges_per_gpu'), cfg.get('levels')),
                      **{k: cfg.get(k) for k in ('vertices_per_gpu', 'edges_per_gpu', 'levels', 'parallelism', 'hip_graph', 'crops_per_step')}}
    r = out.get('roofline') or {}
    line['roofline'] = {k: _num(r.get(k)) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_us', 'algorithmic_bytes',
                                                    'traffic', 'traffic_over_algorithmic')}
    c = out.get('cpu_baseline')
    if c:
        line['cpu_baseline'] = {k: _num(c.get(k)) for k in ('value', 'unit', 'cores', 'kind', 'cpu_model', 'sample_vertices')}
        line['cpu_baseline']['sample'] = 'CPU oracle fwd+loss+bwd, %s-vertex mesh, warm-up + median of %d passes' % (
            c.get('sample_vertices'), len(c.get('passes_s') or []))
    else:
        line['cpu_baseline'] = None

    def leaf(key, sub):
        v = out.get(key)
        return _num(v.get(sub)) if isinstance(v, dict) else None
    line['gemm_precision'] = '%s/%s' % ((out.get('gemm_precision') or {}).get('fwd'), (out.get('gemm_precision') or {}).get('bwd'))
    line['exact_fp32_ms_per_step'] = leaf('exact_fp32', 'ms_per_step')
    line['hbm_honest_frac'] = leaf('hbm_honest', 'frac_of_hbm_peak')
    line['hbm_honest_traffic_over_algorithmic'] = leaf('hbm_honest', 'traffic_over_algorithmic')
    line['scatter_add_frac'] = leaf('scatter_add', 'frac_of_hbm_peak')
    line['roofline_irregular_frac'] = leaf('roofline_irregular', 'frac')
    line['gpu_idle_ms'] = leaf('gpu_idle', 'idle_ms_per_step')
    line['fwd_loss_bwd_only_ms'] = leaf('fwd_loss_bwd_only', 'ms_per_step')
    line['inference_ms'] = leaf('inference', 'ms')
    line['vertex_locality_ms'] = leaf('vertex_locality', 'ms_per_step')
    lf = (out.get('loader_fed') or {}).get('graph_and_plan_resident')
    line['loader_fed_ms'] = _num(lf.get('ms_per_step')) if isinstance(lf, dict) else None
    line['gemm_ms_per_step_standalone'] = leaf('gemm', 'ms_per_step')
    line['gemm_frac_of_roofline'] = leaf('gemm', 'time_weighted_frac_of_roofline')
    line['edge_stage_ms_per_step'] = _num(out.get('edge_stage_ms_per_step'))
    line['host_enqueue_ms_per_step'] = _num(out.get('host_enqueue_ms_per_step'))
    line['loss'] = _num(out.get('loss'))
    d = out.get('distributed') or {}
    line['replicas_bit_identical'] = d.get('replicas_bit_identical')
    line['ms_per_step_per_rank'] = [_num(v, 5) for v in (d.get('ms_per_step_per_rank') or [])][:16]
    line['allreduce_us_mean'] = _num((d.get('allreduce_us') or {}).get('mean')) if d.get('allreduce_us') else None
    line['detail'] = detail_path
    text = json.dumps(line)
    assert len(text) < COMPACT_LIMIT, 'compact bench line grew to %d bytes' % len(text)
    return text


def write_detail(out, path):
    """The full detail object -> `path` (JSON, one object).  Falls back to the temp dir when the target cannot be written."""
    import tempfile
    for cand in (path, os.path.join(tempfile.gettempdir(), 'stin_bench_detail_%d.json' % os.getpid())):
        try:
            os.makedirs(os.path.dirname(os.path.abspath(cand)), exist_ok=True)
            with open(cand, 'w') as f:
                json.dump(out, f)
                f.write('\n')
            return cand
        except OSError:
            continue
    return None


def edge_bytes(kernel, n, e, h):
    """Algorithmic bytes per launch: gathered rows are charged once
    per edge (no cache credit), s = 4 (fp32) or 2 (bf16 storage) bytes per element, int32 indices."""
    idx = 4 * e + 4 * (n + 1)
    s = 2 if kernel.endswith('_bf16') else 4
    kernel = kernel.replace('_bf16', '_f32')
    if kernel == 'stin_edge_relu_mean_fwd_f32':      # gather B per edge, read A, write h
        return (e * h + 2 * n * h) * s + idx
    if kernel == 'stin_edge_relu_mean_bwd_dst_f32':  # gather B per edge, read A and G, write dA
        return (e * h + 3 * n * h) * s + idx
    if kernel == 'stin_edge_relu_mean_bwd_src_f32':  # gather A and G per edge, read B, write dB, inv_deg per edge
        return (2 * e * h + 2 * n * h) * s + idx + 4 * e
    if kernel == 'stin_edge_relu_mean_bwd_dst_mask_f32':  # stream the H-bit masks of the in-edges, read G, write dA
        return e * h // 8 + 2 * n * h * s + 4 * (n + 1)
    if kernel == 'stin_edge_relu_mean_bwd_src_mask_f32':  # gather G per edge + its mask words, col/xslot/inv_deg, write dB
        return e * h * s + e * h // 8 + 12 * e + n * h * s + 4 * (n + 1)
    if kernel == 'stin_edge_relu_mean_bwd_mask_f32':      # both of the above in one launch
        sfx = '_bf16' if s == 2 else '_f32'
        return (edge_bytes('stin_edge_relu_mean_bwd_dst_mask' + sfx, n, e, h) +
                edge_bytes('stin_edge_relu_mean_bwd_src_mask' + sfx, n, e, h))
    if kernel == 'stin_edge_relu_mean_fwd_ti_f32':        # compact trans-inv: the row's own B row instead of its A row - same bytes
        return edge_bytes('stin_edge_relu_mean_fwd_f32', n, e, h)
    if kernel == 'stin_edge_relu_mean_bwd_mask_ti_f32':   # both halves per row, ONE output row D = dB - dA (+ the tiny column partials)
        return edge_bytes('stin_edge_relu_mean_bwd_mask_f32', n, e, h) - n * h * s
    raise KeyError(kernel)


def pmc_traffic_bytes(kernel, n, e, h):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json, collected at
    the headline level-0 shape with profiles/pmc_kernels.py; FETCH_SIZE doubled per the gfx950 correction)."""
    import glob
    if (n, e) != (200704, 1200642):
        return None
    elem = '__bf16' if kernel.endswith('_bf16') else 'float'
    kernel = kernel.replace('_bf16', '_f32')
    short = {'stin_edge_relu_mean_fwd_f32': 'k_edge_fwd', 'stin_edge_relu_mean_bwd_dst_f32': 'k_edge_bwd_dst',
             'stin_edge_relu_mean_bwd_src_f32': 'k_edge_bwd_src',
             'stin_edge_relu_mean_bwd_dst_mask_f32': 'k_edge_bwd_dst_mask',
             'stin_edge_relu_mean_bwd_src_mask_f32': 'k_edge_bwd_src_mask',
             'stin_edge_relu_mean_bwd_mask_f32': 'k_edge_bwd_mask_pair', 'stin_edge_relu_mean_fwd_ti_f32': 'k_edge_fwd_ti',
             'stin_edge_relu_mean_bwd_mask_ti_f32': 'k_edge_bwd_mask_ti'}[kernel]
    if elem == 'float':                                     # Lane<G, VPL>: 4 channels per lane
        c4 = h // 4
        g = 1
        while g < c4 and g < 64:
            g *= 2
        prefix = '%s<float, %d, %d,' % (short, g, (c4 + g - 1) // g)
    else:                                                   # bf16 rows: 8 channels per lane (k_*8 kernels)
        g = min(64, h // 8)
        prefix = '%s8<%d, %d,' % (short, g, h // (8 * g))
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return None
    table = json.load(open(files[-1]))
    hits = [v for k, v in table.items() if k.startswith(prefix) or k.startswith(prefix.replace('k_edge_fwd<', 'k_edge_fwd_exact<'))]
    return (hits[0].get('fabric_MB_per_launch') or hits[0].get('hbm_MB_per_launch')) * 1e6 if hits else None     # (round <= 3 files: hbm_* keys)


def measure_fabric_traffic(kernel_prefix='k_edge_fwd_exact<float, 32, 1, 6>'):
    """LIVE fabric traffic of the level-0 forward edge kernel (round 4: `roofline.traffic`; round 5: also `hbm_honest.traffic`): two
    child processes - `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (they do not fit one TCC pass) with `--kernel-trace` only -
    over profiles/pmc_kernels.py in PMC_LIVE mode: 5 launches on the headline mesh, then 4 at the 1 M-vertex / 6 M-edge size, told
    apart by launch order; bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB per the gfx950 corrections of MI355X_MICROARCH.md.  Children,
    not an exec: this process keeps its GPU context.  -> ({'headline': bytes per launch, 'n1m': bytes per launch}, note) or (None, reason)."""
    import csv
    import glob
    import re
    import shutil
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not on PATH'
    vals = {}
    tmp = tempfile.mkdtemp(prefix='stin_pmc_', dir='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, counter)
            env = dict(os.environ, TMPDIR='/tmp', PMC_LIVE='1')
            r = subprocess.run(['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'run', '--',
                                sys.executable, os.path.join(ROOT, 'profiles', 'pmc_kernels.py')], cwd='/tmp', env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                return None, 'rocprofv3 --pmc %s pass failed (rc %d)' % (counter, r.returncode)
            acc = []
            for row in csv.DictReader(open(files[0])):
                if row.get('Counter_Name') != counter:
                    continue
                name = re.sub(r'\(anonymous namespace\)::|^void ', '', row['Kernel_Name'])
                if name.startswith(kernel_prefix):
                    acc.append((int(row.get('Dispatch_Id', len(acc))), float(row['Counter_Value'])))
            acc = [v for _, v in sorted(acc)]
            if len(acc) < 6:
                return None, 'expected 5 + 4 launches of %s in the %s pass, found %d' % (kernel_prefix, counter, len(acc))
            thr = (min(acc) * max(acc)) ** 0.5                               # the 1 M-vertex launches move ~5x the bytes
            small, large = [v for v in acc if v < thr], [v for v in acc if v >= thr]
            if len(small) < 3 or len(large) < 3:
                return None, 'could not separate the two sizes in the %s pass (%d / %d launches)' % (counter, len(small), len(large))
            vals[counter] = (sum(small[-3:]) / 3.0, sum(large[-3:]) / 3.0)   # mean of the last 3 launches at each size
    except Exception as exc:                                # noqa: BLE001 - a secondary leg must not lose the line
        return None, '%s: %s' % (type(exc).__name__, exc)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {k: (2.0 * vals['FETCH_SIZE'][i] + vals['WRITE_SIZE'][i]) * 1024.0 for i, k in enumerate(('headline', 'n1m'))}, 'measured in this run'


def scatter_add_standalone(device, n=200_000, e=1_200_000, c=64, iters=30):
    """The standalone scatter-add of BASELINE.md §4: src[E, C] -> out[N, C], index in arbitrary edge order."""
    from surface_texture_inpainting_net_amd import functional as SF
    from surface_texture_inpainting_net_amd.plan import build_csr
    g = torch.Generator().manual_seed(0)
    index = torch.randint(0, n, (e,), generator=g).to(device)
    src = torch.randn(e, c, device=device)
    bad = torch.zeros(1, dtype=torch.int32, device=device)
    csr = build_csr(index, None, n, e, bad, want_perm=True)
    for _ in range(3):
        SF.segment_sum(src, csr.rowptr, csr.perm, n)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        SF.segment_sum(src, csr.rowptr, csr.perm, n)
    b.record()
    torch.cuda.synchronize()
    dt = a.elapsed_time(b) * 1e-3 / iters
    nbytes = e * c * 4 + n * c * 4 + 4 * e + 4 * (n + 1)
    return {'E': e, 'N': n, 'C': c, 'us': dt * 1e6, 'algorithmic_MB': nbytes / 1e6,
            'GBps': nbytes / dt / 1e9, 'frac_of_hbm_peak': nbytes / dt / 1e9 / HBM_PEAK_GBS}


def cpu_baseline(n0_target, levels, seed, headline_mesh=True):
    """The CPU oracle (op-for-op unfused PyG form) timed on this box's host cores, SURVEY 8(d) protocol: one warm-up pass AT
    SIZE, then the median of 3 timed passes.  Round 5: `value` is measured IN THIS RUN on the SAME synthetic mesh the GPU step
    ran on (200 704 vertices: warm-up + 3 passes of ~20 s each at the probe-best thread count, ~85 s in all - the bounded sample
    of the contract is three passes of the headline workload itself); the ~40 k-vertex sample of rounds 1-4 (one pass ~4 s)
    rides along as `quick_sample` and is all that runs with --quick-cpu-baseline.  The thread count is what a short probe over
    {8, 16, 32, 64} found fastest (torch's CPU index / scatter ops slow down badly when oversubscribed across a 256-thread
    host) - reported as `cores`; `all_cores` is the same protocol with torch.set_num_threads(os.cpu_count()) (warm-up + median
    of up to 3, cut short when a pass exceeds its budget)."""
    import statistics
    from oracle import stin_oracle
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    torch.manual_seed(49)
    net = stin_oracle.define_G(**CONFIG_3D)

    def run(sample):
        net.zero_grad(set_to_none=True)
        t = time.perf_counter()
        loss = stin_oracle.compute_loss(stin_oracle.graph_forward(net, sample), sample.color, sample.mask)
        loss.backward()
        return time.perf_counter() - t

    probe = make_synthetic_mesh(10_000, levels, seed=seed)
    ncpu = os.cpu_count() or 1
    best_t, best_threads, probe_log = None, None, {}
    for th in [t for t in (8, 16, 32, 64) if t <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        run(probe)                               # warm-up (allocator, thread pool)
        t = run(probe)
        probe_log[th] = probe.x.shape[0] / t
        if best_t is None or t < best_t:
            best_t, best_threads = t, th
        if t > 8.0:                              # keep the probe itself bounded
            break
    per_vertex = best_t / probe.x.shape[0]
    # sample size: one pass ~4 s.  The per-vertex cost grows with the mesh (caches: 34 k vertices/s at 10 k vertices, 10 k at
    # 200 k on the EPYC 9575F boxes), hence the factor 3 on the probe's rate.
    n0 = int(min(n0_target, max(10_000, 4.0 / (3.0 * per_vertex))))
    sample = probe if n0 <= probe.x.shape[0] else make_synthetic_mesh(n0, levels, seed=seed)
    nv = sample.x.shape[0]

    def protocol(smp, threads, budget_s):
        """-> (warm-up seconds, [timed passes]): up to 3 passes, none started once the budget would be exceeded."""
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        warm = run(smp)
        ts = []
        while len(ts) < 3 and time.perf_counter() - t0 + max([warm] + ts) <= budget_s:
            ts.append(run(smp))
        return warm, ts

    warm, ts = protocol(sample, best_threads, 30.0)
    med = statistics.median(ts) if ts else warm
    out = {'value': nv / med, 'unit': 'vertices/s', 'cores': best_threads, 'kind': 'port', 'cpu_model': _cpu_model(),
           'host_threads': ncpu, 'sample_vertices': nv, 'passes_s': [round(t, 3) for t in ts], 'warmup_s': round(warm, 3),
           'thread_probe_vertices_per_s': {str(k): round(v, 1) for k, v in probe_log.items()}}
    if ncpu != best_threads:
        # SURVEY 8(d) names torch.set_num_threads(os.cpu_count()).  On the 256-thread hosts that is ~100x SLOWER than the
        # probe-best count (oversubscribed scatter / index kernels: round 3 spent 132 s in ONE un-warmed pass on the 10 k-vertex
        # probe mesh and printed a figure from it).  The leg therefore runs only where the probe says it can finish: the
        # largest probed thread count within 4x of the best rate and the host no more than twice as wide as that probe;
        # otherwise it is reported as skipped with the probe evidence, and a value is never derived from an empty pass list.
        th_max = max(probe_log)
        if probe_log[th_max] * 4.0 >= probe_log[best_threads] and ncpu <= 2 * th_max:
            warm_a, ts_a = protocol(probe, ncpu, 15.0)
            out['all_cores'] = {'cores': ncpu, 'sample_vertices': probe.x.shape[0], 'unit': 'vertices/s',
                                'passes_s': [round(t, 3) for t in ts_a], 'warmup_s': round(warm_a, 3),
                                'value': probe.x.shape[0] / statistics.median(ts_a) if ts_a else None,
                                'same_mesh_at_probe_best_threads_vertices_per_s': round(probe_log[best_threads], 1)}
        else:
            out['all_cores'] = {'cores': ncpu, 'skipped': 'thread probe: %d threads already run at %.2f of the %d-thread rate on the '
                                '10 k-vertex probe mesh; %d threads would only thrash the intra-op pool (round 3: 75.8 vertices/s '
                                'from one 132 s pass)' % (th_max, probe_log[th_max] / probe_log[best_threads], best_threads, ncpu)}
    torch.set_num_threads(best_threads)
    what = ('fwd+loss+bwd of the CPU oracle (unfused PyG-form restatement, torch %s CPU, fp32) on a synthetic %d-vertex %d-level mesh: '
            '1 warm-up at size + median of %d passes (%.2f s) with %d of %d host threads of %s (probe-best)')
    out['sample'] = what % (torch.__version__, nv, levels, len(ts), med, best_threads, ncpu, _cpu_model())
    if headline_mesh and n0_target > nv:
        # the headline mesh itself (same generator call as the GPU step's scene: make_synthetic_mesh(vertices, levels, seed=0))
        quick = {k: out[k] for k in ('value', 'sample_vertices', 'passes_s', 'warmup_s', 'sample')}
        full = make_synthetic_mesh(n0_target, levels, seed=seed)
        warm_f, ts_f = protocol(full, best_threads, 240.0)
        med_f = statistics.median(ts_f) if ts_f else warm_f
        nf = full.x.shape[0]
        out.update({'value': nf / med_f, 'sample_vertices': nf, 'edges': int(full.edge_index.shape[1]),
                    'passes_s': [round(t, 3) for t in ts_f], 'warmup_s': round(warm_f, 3), 'quick_sample': quick,
                    'sample': what % (torch.__version__, nf, levels, len(ts_f), med_f, best_threads, ncpu, _cpu_model()) +
                              ' - the SAME mesh the GPU step ran on, timed in this run; quick_sample = the bounded sample of rounds 1-4'})
    out['sample'] += ('; all_cores = the same protocol with all %d threads on the 10 k-vertex probe mesh, run only when the probe says '
                      'it can finish' % ncpu)
    return out


def exact_fp32_companion(args):
    """What the SAME step costs with exact-fp32 GEMMs (v_mfma_f32_32x32x2_f32 on every product, STIN_GEMM_FWD=0 STIN_GEMM_BWD=0 -
    the reference's arithmetic is plain fp32 addmm): a child process of this script, started after the timed region (a child, not an
    exec: this process keeps its GPU context; the precision switches are read at import), short loop, no secondary legs."""
    env = dict(os.environ, STIN_GEMM_FWD='0', STIN_GEMM_BWD='0')
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', '15', '--warmup', '5', '--vertices', str(args.vertices),
           '--levels', str(args.levels), '--no-secondary', '--no-cpu-baseline', '--no-live-traffic',
           '--detail', os.path.splitext(args.detail)[0] + '_exact_fp32.json']
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
        line = [ln for ln in r.stdout.decode(errors='replace').splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not line:
            return {'error': 'child run failed (rc %d)' % r.returncode}
        c = json.loads(line[-1])
        return {'ms_per_step': c['ms_per_step'], 'vertices_per_s': c['value'],
                'gemm_precision': {'fwd/bwd': c['gemm_precision'],
                                   'note': 'v_mfma_f32_32x32x2_f32 on unsplit fp32 operands, fp32 accumulate'},
                'loss': c['loss'], 'steps': c['steps'], 'warmup': c['warmup'],
                'note': 'the same step, mesh and seeds with EXACT fp32 matrix-core products in every GEMM, forward and backward '
                        '(STIN_GEMM_FWD=0 STIN_GEMM_BWD=0), measured in this run by a child process after the timed region; the headline '
                        '`value` uses the split-16-bit products named in gemm_precision, which meet the stated fp32 tolerances '
                        '(dtype_tolerance) at this size'}
    except Exception as exc:                                # noqa: BLE001 - a secondary leg must not lose the line
        return {'error': '%s: %s' % (type(exc).__name__, exc)}


def backlogged_step_ms(one_step, steps=10):
    """GPU time of a step when the host is out of the way: a ~60 ms sleep kernel holds the compute stream while the host enqueues
    `steps` steps behind it, HIP events around those steps on the same stream.  ms_per_step (wall, host in the loop) minus this =
    the GPU idle time per step that a profiler-free run leaves at launch / step boundaries."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cyc = 20_000_000
    a.record()
    torch.cuda._sleep(cyc)
    b.record()
    torch.cuda.synchronize()
    per_ms = cyc / max(a.elapsed_time(b), 1e-3)
    one_step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(per_ms * 60.0))
    e0.record()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    host_ms = (time.perf_counter() - t0) * 1e3
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, host_ms / steps


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def hbm_honest_edge_kernel(device, n=1_000_000, e=6_000_000, h=128, iters=10, live_bytes=None):
    """The level-0 forward edge kernel at the 1 M-vertex / 6 M-edge size of BASELINE config 5: its gathered operand B is
    n*h*4 = 512 MB, twice the 256 MB Infinity Cache, so this figure is an HBM figure (at 200 k vertices B is 102 MB and
    lives in the Infinity Cache - the headline `roofline` is the algorithmic-byte convention of SURVEY §8d)."""
    from surface_texture_inpainting_net_amd import functional as SF
    from surface_texture_inpainting_net_amd.plan import EdgeSet
    g = torch.Generator().manual_seed(1)
    ei = torch.randint(0, n, (2, e), generator=g).to(device)
    bad = torch.zeros(1, dtype=torch.int32, device=device)
    edges = EdgeSet(ei, n, bad)
    Y = torch.randn(n, 2 * h, device=device)
    out = torch.empty(n, h + 4, device=device)
    mask = torch.empty(e * (h // 32), dtype=torch.int32, device=device)
    for _ in range(2):
        SF.edge_relu_mean_fwd(Y[:, :h], Y[:, h:], edges.by_dst, out, indicator=True, mask=mask)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        SF.edge_relu_mean_fwd(Y[:, :h], Y[:, h:], edges.by_dst, out, indicator=True, mask=mask)
    b.record()
    torch.cuda.synchronize()
    dt = a.elapsed_time(b) * 1e-3 / iters
    nbytes = edge_bytes('stin_edge_relu_mean_fwd_f32', n, e, h)
    out = {'kernel': 'stin_edge_relu_mean_fwd_f32[N=%d,E=%d,H=%d]' % (n, e, h), 'us': dt * 1e6, 'algorithmic_MB': nbytes / 1e6,
           'GBps': nbytes / dt / 1e9, 'frac_of_hbm_peak': nbytes / dt / 1e9 / HBM_PEAK_GBS,
           'note': 'gathered operand 512 MB > 256 MB Infinity Cache: served by HBM (random graph, fp32 rows)'}
    if live_bytes is not None and (n, e, h) == (1_000_000, 6_000_000, 128):
        out['traffic'] = live_bytes
        out['traffic_over_algorithmic'] = live_bytes / (nbytes + e * h / 8.0)
        out['traffic_unit'] = ('FABRIC bytes per launch ((2*FETCH_SIZE + WRITE_SIZE) KiB), MEASURED in this run: the same two rocprofv3 --pmc child '
                               'passes as roofline.traffic (profiles/pmc_kernels.py PMC_LIVE=1: this kernel at this shape, mean of the last 3 '
                               'launches); the counters sit on the L2\'s memory side and include Infinity-Cache hits, so the HBM-served share '
                               'cannot be separated with them: with a 512 MB gathered operand, random row order and a 256 MB cache at most half of '
                               'the gathered rows can hit; fabric bytes = %.3f x (algorithmic + mask) bytes, i.e. no reuse is captured in L2'
                               % (live_bytes / (nbytes + e * h / 8.0)))
    else:
        out['traffic'] = None
        out['traffic_unit'] = 'not measured in this run (live PMC passes skipped or unavailable)'
    return out


def irregular_edge_kernel(device, n0=200_000, h=128, iters=20):
    """The level-0 forward edge kernel on an IRREGULAR 200 k-vertex mesh (Delaunay triangulation of random points: vertex
    degrees 3 ... ~18, mean 6, sigma 1.3 - the valence spread of a QEM-decimated scan) beside the 6-regular jittered grid
    of the headline: what the degree spread alone costs the one-lane-group-per-row kernels."""
    from surface_texture_inpainting_net_amd import functional as SF
    from surface_texture_inpainting_net_amd.plan import EdgeSet
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    s = make_synthetic_mesh(n0, 1, seed=0, dilations=(), irregular=True)
    ei = s.edge_index.to(device)
    n, e = s.x.shape[0], ei.shape[1]
    deg = torch.bincount(s.edge_index[1], minlength=n)
    bad = torch.zeros(1, dtype=torch.int32, device=device)
    edges = EdgeSet(ei, n, bad)
    Y = torch.randn(n, 2 * h, device=device)
    out = torch.empty(n, h + 4, device=device)
    mask = torch.empty(e * (h // 32), dtype=torch.int32, device=device)
    g = torch.randn(n, h, device=device)
    dY = torch.empty(n, 2 * h, device=device)

    def timed(fn):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / iters

    tf = timed(lambda: SF.edge_relu_mean_fwd(Y[:, :h], Y[:, h:], edges.by_dst, out, indicator=True, mask=mask))
    tb = timed(lambda: SF.edge_relu_mean_bwd_mask(g, mask, edges, dY[:, :h], dY[:, h:]))
    bf, bb = edge_bytes('stin_edge_relu_mean_fwd_f32', n, e, h), edge_bytes('stin_edge_relu_mean_bwd_mask_f32', n, e, h)
    return {'kernel': 'stin_edge_relu_mean_fwd_f32[N=%d,E=%d,H=%d]' % (n, e, h), 'bound': 'hbm', 'avg_us': tf * 1e6,
            'algorithmic_bytes': bf, 'achieved': bf / tf / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': bf / tf / 1e9 / HBM_PEAK_GBS,
            'backward': {'kernel': 'stin_edge_relu_mean_bwd_mask_f32', 'avg_us': tb * 1e6, 'GBps': bb / tb / 1e9,
                         'frac': bb / tb / 1e9 / HBM_PEAK_GBS},
            'in_degree': {'min': int(deg.min()), 'max': int(deg.max()), 'mean': float(deg.float().mean()),
                          'std': float(deg.float().std())},
            'note': 'stand-alone, fp32 rows, Delaunay mesh (synthetic.make_synthetic_mesh(irregular=True)); the headline mesh is 6-regular'}


def loader_fed_step(device, net, step, vertices, levels, scenes=3, epochs=3):
    """PCIe-inclusive secondary figure (never `value`): the same training step fed by loader.SceneLoader from HOST-resident
    scenes - every step uploads a different scene (features + int64 index tensors through the pinned staging ring on the copy
    stream) and builds its CSR plan; `resident` = the same loop with the graph part + plan of a revisited scene kept in HBM
    (only x / color / mask cross PCIe)."""
    from surface_texture_inpainting_net_amd.loader import SceneLoader
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    items = [make_synthetic_mesh(vertices, levels, seed=1000 + i) for i in range(scenes)]
    nv = sum(int(it.x.shape[0]) for it in items)
    out = {}
    for name, cache_bytes, loc in (('fresh_scene_every_step', 0, True), ('graph_and_plan_resident', 32 << 30, True),
                                   ('graph_and_plan_resident_file_order', 32 << 30, False)):
        ld = SceneLoader(items, device, shuffle=False, cache_bytes=cache_bytes, model=net, end_level=levels, locality_order=loc)
        for smp in ld.epoch(0):                                 # untimed: pinned ring, allocator pools (and the resident cache)
            step(smp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e in range(epochs):
            for smp in ld.epoch(1 + e):
                step(smp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {'ms_per_step': dt / (epochs * scenes) * 1e3, 'vertices_per_s': nv * epochs / dt}
    step.finish()
    out['note'] = ('%d host-resident synthetic scenes of the headline size through loader.SceneLoader (worker threads, pinned '
                   'staging ring, copy stream), %d epochs after one untimed epoch; fresh = upload + plan build every step, '
                   'resident = graph tensors + plan cached in HBM, features uploaded; the resident plan is built with the vertices '
                   'renumbered by locality (the loader default, paid once per scene; colours come back in the scene\'s vertex order), '
                   '_file_order = the same with locality_order=False' % (scenes, epochs))
    return out


def rccl_debug_setup(rank):
    """Before the process group exists: have RCCL log its setup (INFO: init, graph, tuning) into a per-process file, so that
    the line can say which algorithm / protocol / channel count the gradient all-reduce ran with.
    Nothing goes to stdout (ONE JSON line).  -> the log path of this process."""
    import tempfile
    os.environ.setdefault('NCCL_DEBUG', 'INFO')
    os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH,TUNING,ENV')
    path = os.environ.get('NCCL_DEBUG_FILE')
    if path is None:
        path = os.path.join(tempfile.gettempdir(), 'stin_rccl_%d_rank%d.log' % (os.getppid(), rank))
        os.environ['NCCL_DEBUG_FILE'] = path
    return path


def rccl_debug_parse(path, nbytes):
    """What RCCL said about itself: version, channels, rings / trees, and the (algorithm, protocol) its tuner picked for the
    gradient bucket's size.  Tolerant by design (the wording differs between RCCL releases): every figure is optional and
    the matched lines ride along verbatim."""
    import re
    out = {'log': path, 'NCCL_ALGO': os.environ.get('NCCL_ALGO'), 'NCCL_PROTO': os.environ.get('NCCL_PROTO'),
           'NCCL_MIN_NCHANNELS': os.environ.get('NCCL_MIN_NCHANNELS'), 'NCCL_MAX_NCHANNELS': os.environ.get('NCCL_MAX_NCHANNELS')}
    try:
        text = open(path, errors='replace').read()
    except OSError:
        out['note'] = 'no RCCL debug log found'
        return out
    lines = text.splitlines()
    keep = []

    def first(pattern, flags=re.I):
        for ln in lines:
            m = re.search(pattern, ln, flags)
            if m:
                keep.append(ln.strip()[:240])
                return m
        return None

    m = first(r'(?:RCCL|NCCL) version ([^\s]+)')
    out['version'] = m.group(1) if m else None
    m = first(r'(\d+) coll channels')
    out['coll_channels'] = int(m.group(1)) if m else None
    chans = [ln for ln in lines if re.search(r'Channel \d+/\d+', ln)]
    if chans:
        m = re.search(r'Channel \d+/(\d+)', chans[0])
        out['ring_channels'] = int(m.group(1))
        keep.append(chans[0].strip()[:240])
    out['rings_connected'] = first(r'Connected all rings') is not None
    out['trees_connected'] = first(r'Connected all trees') is not None
    # tuner decisions: "<bytes> Bytes -> Algo <a> proto <p> time <t>"; report the one closest to the bucket size
    algo_names = {0: 'Tree', 1: 'Ring', 2: 'CollNetDirect', 3: 'CollNetChain', 4: 'NVLS', 5: 'NVLSTree'}
    proto_names = {0: 'LL', 1: 'LL128', 2: 'Simple'}
    best = None
    for ln in lines:
        m = re.search(r'(\d+) Bytes -> Algo (\d+) proto (\d+)', ln)
        if m:
            b = int(m.group(1))
            if best is None or abs(b - nbytes) < abs(best[0] - nbytes):
                best = (b, int(m.group(2)), int(m.group(3)), ln.strip()[:240])
    if best is not None:
        out['allreduce_tuning'] = {'bytes': best[0], 'algo': algo_names.get(best[1], best[1]), 'proto': proto_names.get(best[2], best[2])}
        keep.append(best[3])
    out['lines'] = keep[:12]
    out['log_lines_total'] = len(lines)
    return out


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks as children of a
    torch.distributed.run process.  This parent has not touched the GPU (counting devices does not initialise HIP) and
    never execs: it waits for the launcher and exits with its status, so one failed rank fails the run."""
    ndev = torch.cuda.device_count()
    if args.backend == 'nccl' and ndev < args.gpus:
        sys.stderr.write('bench.py: --gpus %d but only %d GPU(s) visible; RCCL needs one device per rank '
                         '(use --backend gloo to smoke-test the multi-rank path on fewer GPUs)\n' % (args.gpus, ndev))
        sys.exit(2)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def pin_rank_to_cores(local_rank, local_world):
    """One contiguous share of the host's cores per rank (8 ranks enqueue ~4-5 ms of launches per step each; left
    floating they migrate across sockets and fight over cores).  -> the cores this rank may run on."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        share = max(1, len(cores) // local_world)
        mine = cores[local_rank * share:(local_rank + 1) * share] or cores
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), 16)))
        return len(mine)
    except (AttributeError, OSError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--vertices', type=int, default=200_000)
    ap.add_argument('--levels', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--quick-cpu-baseline', action='store_true', help='cpu_baseline on the bounded ~40 k-vertex sample only (rounds 1-4); the '
                    'default also times the CPU oracle on the headline mesh itself (~85 s)')
    ap.add_argument('--no-exact-fp32', action='store_true', help='skip the exact-fp32-GEMM companion figure (a child run of ~25 s)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' only to smoke-test "
                    "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument('--time-gemms', action='store_true', help='bracket every MFMA GEMM launch INSIDE the timed region too '
                    '(the default line times the GEMMs in a separate pass after it)')
    ap.add_argument('--graph', action='store_true', help='TrainStep(graph=True): plan build + fwd + loss + bwd captured into one HIP '
                    'graph and replayed (for launch-bound configurations; the per-kernel HIP-event brackets are not available)')
    ap.add_argument('--no-prefetch-plan', action='store_true', help='build each step\'s CSR plan on the compute stream at first '
                    'use instead of one step ahead on the plan side streams')
    ap.add_argument('--cache-plan', action='store_true', help='reuse the CSR plan across steps (NOT the headline)')
    ap.add_argument('--crops', type=int, default=0,
                    help='BASELINE config 3: a collated batch of this many unequal crops (12-28k vertices each) per step '
                         'instead of one scene (NOT the headline); combine with --levels 4 --dtype bf16')
    ap.add_argument('--unequal-scenes', action='store_true', help='config 4 as the reference trains it: rank r gets a scene of '
                    '150 000 + r * 50 000 / (N - 1) vertices instead of N equal ones (NOT the headline; reports the straggler figure)')
    ap.add_argument('--no-live-traffic', action='store_true', help='skip the two rocprofv3 --pmc child passes that measure roofline.traffic '
                    'and hbm_honest.traffic live (about 40 s); both are null then')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the passes after the timed region (fwd+loss+bwd-only loop, GEMM table, standalone kernels, CPU '
                         'baseline) - profiling runs: keeps the kernel mix = the step')
    ap.add_argument('--coherent-order', action='store_true', help='experiment: keep the grid (row-major) vertex order instead of the '
                    'random permutation the synthetic meshes get by default (an upper bound on what vertex locality is worth)')
    ap.add_argument('--morton-order', action='store_true', help='experiment: renumber the vertices of every level by locality on the host '
                    'before the run (synthetic.renumber_by_locality)')
    ap.add_argument('--irregular', action='store_true', help='the whole step on an irregular (Delaunay) mesh of the same size '
                    'instead of the 6-regular jittered grid (NOT the headline)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="activation storage: f32 = the headline (reference numerics); bf16 = the build's "
                         "mixed-precision mode of BASELINE configs 3/5 (NOT the headline, stated tolerance)")
    ap.add_argument('--detail', default=os.path.join(ROOT, 'gpurun_out', 'bench_detail.json'),
                    help='file the FULL detail object goes to (stdout carries only the compact line)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)                                     # does not return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    if world != args.gpus:
        sys.stderr.write('bench.py: WORLD_SIZE=%d but --gpus %d - launch with --nproc-per-node == --gpus\n' % (world, args.gpus))
        sys.exit(2)
    cores_per_rank = pin_rank_to_cores(local_rank, local_world) if world > 1 else None
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    rccl_log = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            rccl_log = rccl_debug_setup(rank)
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from surface_texture_inpainting_net_amd import _lib
    from surface_texture_inpainting_net_amd import functional as SF
    from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    from surface_texture_inpainting_net_amd.train_step import TrainStep, replicas_identical
    _lib.load()

    torch.manual_seed(49)                                   # reference config seed; identical replicas
    cfg = dict(CONFIG_3D)
    if args.levels != 3:                                    # other hierarchy depths (configs 3 and 5): n_levels = levels - 1
        cfg['n_levels'] = args.levels - 1
    net = S.define_G(**cfg).to(device)
    if args.dtype == 'bf16':
        net.set_activation_dtype(torch.bfloat16)
    step = TrainStep(net, lr=7e-5, amsgrad=True, time_allreduce=world > 1, graph=args.graph, freeze_gc=True)
    if args.crops > 0:
        from surface_texture_inpainting_net_amd.data import collate
        sizes = [12_000 + (16_000 * i) // max(args.crops - 1, 1) for i in range(args.crops)]
        sample = collate([make_synthetic_mesh(n, args.levels, seed=100 * rank + i) for i, n in enumerate(sizes)]).to(device)
    else:
        nv_rank = args.vertices
        if args.unequal_scenes and world > 1:
            nv_rank = 150_000 + (50_000 * rank) // (world - 1)
        sample = make_synthetic_mesh(nv_rank, args.levels, seed=rank, irregular=args.irregular,
                                     permute=not args.coherent_order)               # one scene per rank
        if args.morton_order:
            from surface_texture_inpainting_net_amd.synthetic import renumber_by_locality
            sample = renumber_by_locality(sample)[0]
        sample = sample.to(device)
    n0 = sample.x.shape[0]
    e0 = sample.edge_index.shape[1]

    pending_plan = [None]

    def one_step(prefetch=True, behind_compute=False):
        if not args.cache_plan:
            # a fresh CSR plan per step (part of the step).  As a loader with one batch of look-ahead does
            # (TrainStep.prefetch), the plan of step k+1 is built on the plan side streams while step k runs; the
            # first step builds its own at first use.  --no-prefetch-plan: every step builds its plan on the compute stream.
            # prefetch=False (the HIP-event-bracketed step): nothing runs on the side streams while kernels are being timed -
            # the step after it then builds its plan at first use on the compute stream.
            sample._plan_cache = pending_plan[0]
            # behind_compute=True (the step AFTER the bracketed one): its look-ahead build starts only when everything enqueued so
            # far has finished on the GPU - at 1 M vertices the host runs a whole step ahead of the GPU, so a build enqueued
            # here would otherwise run beside the bracketed step's kernels (k_count, 1.1 ms at 1 M, beside a level-0 edge
            # launch: 1 290 us instead of 530 in the rocprofv3 trace; which launch it hit changed from run to run)
            pending_plan[0] = None if (args.no_prefetch_plan or args.graph or not prefetch) else \
                net.build_plan(sample, inputs_ready=not behind_compute)
        return step(sample)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    # at least five untimed steps before the timed region whatever W is: the first steps of a process grow the allocator pools,
    # upload constants and (with Python's collector still enabled) pay generation-2 passes - 8.8 instead of 7.5 ms per step
    # over the first 35 steps when measured cold.  Reported as `priming_steps` beside `warmup`.
    priming = max(0, 5 - args.warmup)
    for _ in range(args.warmup + priming):
        one_step()
    gc.collect()
    gc.disable()            # no cyclic-GC pause inside the timed region (collected again right after it)
    fence()
    sfx = '_' + args.dtype
    edge_names = ['stin_edge_relu_mean_fwd' + sfx, 'stin_edge_relu_mean_bwd_dst_f32', 'stin_edge_relu_mean_bwd_src_f32',
                  'stin_edge_relu_mean_bwd_dst_mask' + sfx, 'stin_edge_relu_mean_bwd_src_mask' + sfx,
                  'stin_edge_relu_mean_bwd_mask' + sfx, 'stin_edge_relu_mean_fwd_ti_f32', 'stin_edge_relu_mean_bwd_mask_ti_f32']
    gemm_names = ['stin_gemm_nt' + sfx, 'stin_gemm_tn' + sfx]
    timed = edge_names + (gemm_names if args.time_gemms else [])
    # HIP-event brackets need the per-kernel host path (the whole-block C calls enqueue their kernels natively) and event
    # pairs are not free: bracket the edge launches of the FIRST timed step, the rest runs un-instrumented.
    def eager_step(prefetch=False):
        """one_step() through the per-launch host path even with --graph (HIP-event brackets cannot sit inside a replay).
        prefetch=False: no next-step plan build on the side streams beside the bracketed kernels (round 2's config-5 brackets
        timed that contention instead of the kernel: 997 us bracketed vs 390 us in the rocprof trace)."""
        g, step.graph = step.graph, False
        try:
            return one_step(prefetch=prefetch)
        finally:
            step.graph = g

    # Round 3: the edge-stage brackets sit INSIDE the whole-network call (stin_net_op_t::ev_edge0 / 1), so the bracketed step runs
    # the same fast host path as every other step (the per-kernel path cost ~1.5 ms more for that one step); GEMM brackets
    # (--time-gemms) and models that do not take the whole-network path still use the per-kernel path.
    in_net = not args.time_gemms
    SF.KernelTimer.start(timed, max_records=1_000_000, in_net=in_net)
    eager_step()                                            # one more untimed step: counts the bracketed launches per step
    if in_net and not SF.KernelTimer.records:
        in_net = False
        SF.KernelTimer.stop()
        SF.KernelTimer.start(timed, max_records=1_000_000)
        eager_step()
    per_step = max(1, len(SF.KernelTimer.records))
    SF.KernelTimer.stop()
    fence()
    step.bucket.allreduce_log = []
    if not args.graph:
        # (ONE bracketed step per run: a bracketed step takes the per-kernel host path and costs ~1.5 ms more than a plain one)
        SF.KernelTimer.start(timed, max_records=1_000_000 if args.time_gemms else per_step, in_net=in_net)
    bracketed = SF.KernelTimer.enabled
    t0 = time.perf_counter()
    cpu0 = time.thread_time()
    for it in range(args.steps):
        # the bracketed step (the first timed one) leaves the plan side streams idle: its HIP events time kernels, not contention
        loss = one_step(prefetch=not (SF.KernelTimer.enabled and it == 0), behind_compute=(bracketed and it == 1))
    dt_enqueue = time.perf_counter() - t0                   # host side done (everything enqueued); the GPU may still be running
    dt_cpu = time.thread_time() - cpu0                      # CPU time of the enqueuing thread (the wall figure above also contains
                                                            # waits and, on the shared pool hosts, other tenants' interference)
    fence()
    dt = time.perf_counter() - t0
    if args.graph:                                          # kernel brackets from two eager steps AFTER the timed replays
        SF.KernelTimer.start(timed, max_records=1_000_000, in_net=in_net)
        for _ in range(2):
            eager_step()
        fence()
    ktimes = SF.KernelTimer.stop()
    # the same brackets WITH the next step's plan build co-running on the side streams (what un-bracketed steps experience):
    # reported as `contention`, never as the roofline figure
    ctimes = {}
    if not (args.no_secondary or args.cache_plan or args.no_prefetch_plan or args.graph):
        fence()
        SF.KernelTimer.start(edge_names, max_records=1_000_000, in_net=in_net)
        eager_step(prefetch=True)
        ctimes = SF.KernelTimer.stop()
        fence()
    step.finish()                                           # deferred index checks of the timed steps (all clean)
    allreduce_us = None
    if world > 1 and step.bucket.allreduce_log:
        ts = [a.elapsed_time(b) * 1e3 for a, b in step.bucket.allreduce_log]
        allreduce_us = {'mean': sum(ts) / len(ts), 'min': min(ts), 'max': max(ts), 'bytes': step.bucket.flat.numel() * 4,
                        'overlapped_segments': len(step.bucket.segments or []),
                        'note': 'HIP events around the end-of-backward all-reduce of the flat fp32 gradient bucket (rank 0); '
                                'segments reduced during the backward pass are not inside the bracket'}
    # secondary figure, BASELINE's literal metric definition (fwd + loss + bwd of one scene; CSR plan reused, no
    # all-reduce, no optimizer) - reported beside the headline, never instead of it
    fence()
    t1 = time.perf_counter()
    for _ in range(0 if args.no_secondary else args.steps):
        step.forward_backward(sample)
    fence()
    dt_fb = time.perf_counter() - t1
    # GEMM pass (after the timed region): every MFMA GEMM launch of two more steps bracketed with HIP events; the
    # per-kernel path runs the weight-gradient GEMMs on the compute stream, so these are stand-alone durations
    gtimes = {}
    if args.time_gemms:
        gtimes = {k: v for k, v in ktimes.items() if k[0] in gemm_names}
        gemm_steps = float(args.steps)
    elif not args.no_secondary:
        gemm_steps = 2.0
        SF.KernelTimer.start(gemm_names, max_records=1_000_000)
        for _ in range(int(gemm_steps)):
            eager_step()
        gtimes = SF.KernelTimer.stop()
    gc.enable()
    rank_ms = [dt / args.steps * 1e3]
    # this rank's aggregation figure (the level-0 forward edge kernel = the forward launch with the most algorithmic bytes), for the
    # 1/2/4/8 table row: summed over the ranks below
    my_gbps = 0.0
    fw = [(edge_bytes(name, *tag), sum(ts) / len(ts)) for (name, tag), ts in ktimes.items() if name.startswith('stin_edge_relu_mean_fwd')]
    if fw:
        nb, avg = max(fw)
        my_gbps = nb / avg / 1e9
    sum_gbps = my_gbps
    ar_wait_us = None
    if world > 1 and step.bucket.allreduce_log:
        # every rank's mean time inside the end-of-backward all-reduce bracket: the ranks with the smaller scenes arrive early and
        # WAIT there for the largest one - the straggler effect made visible (the wall times per rank equalise by construction)
        mine = sum(a.elapsed_time(b) * 1e3 for a, b in step.bucket.allreduce_log) / len(step.bucket.allreduce_log)
        w = torch.zeros(world, dtype=torch.float64, device=device)
        w[rank] = mine
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        ar_wait_us = [float(v) for v in w.tolist()]
    if world > 1:
        gb = torch.tensor([my_gbps], dtype=torch.float64, device=device)
        dist.all_reduce(gb, op=dist.ReduceOp.SUM)
        sum_gbps = float(gb.item())
        per_rank = torch.zeros(world, dtype=torch.float64, device=device)
        per_rank[rank] = dt
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        rank_ms = [float(v) / args.steps * 1e3 for v in per_rank.tolist()]
        dt = max(float(v) for v in per_rank.tolist())       # MAX over ranks
        nv = torch.tensor([float(n0), 1.0], dtype=torch.float64, device=device)
        dist.all_reduce(nv, op=dist.ReduceOp.SUM)
        total_vertices, ranks_counted = float(nv[0].item()), int(round(float(nv[1].item())))
        identical = replicas_identical(net)                 # parameters bit-identical across ranks after the run
        if not identical:
            sys.stderr.write('bench.py: replicas diverged (parameters differ across ranks after %d steps)\n' % args.steps)
    else:
        total_vertices, ranks_counted, identical = float(n0), 1, True

    if rank == 0:
        table, gemms = [], []
        for (name, tag), ts in ktimes.items():
            if 'gemm' in name:
                continue
            n, e, h = tag
            nbytes = edge_bytes(name, n, e, h)
            avg = sum(ts) / len(ts)
            table.append({'kernel': name, 'N': n, 'E': e, 'H': h, 'launches': len(ts), 'avg_us': avg * 1e6,
                          'each_us': [round(t * 1e6, 1) for t in ts[:8]], 'total_ms': sum(ts) * 1e3, 'algorithmic_MB': nbytes / 1e6, 'GBps': nbytes / avg / 1e9})
        esz = 2.0 if args.dtype == 'bf16' else 4.0
        for (name, tag), ts in gtimes.items():
            m, nc, k = tag
            avg = sum(ts) / len(ts)
            min_bytes = esz * (m * nc + m * k) + 4.0 * nc * k
            mfma_flops = 2.0 * m * nc * k * (1 if args.dtype == 'bf16' else 3)     # executed on the 16-bit matrix cores
            bound_us = max(min_bytes / (HBM_COPY_GBS * 1e9), mfma_flops / (MFMA_16BIT_PEAK_TF * 1e12)) * 1e6
            gemms.append({'kernel': name, 'M': m, 'Nc': nc, 'K': k, 'launches': len(ts), 'avg_us': avg * 1e6,
                          'total_ms': sum(ts) * 1e3, 'TFLOPs': 2.0 * m * nc * k / avg / 1e12,
                          'mfma_TFLOPs_executed': mfma_flops / avg / 1e12, 'GBps_min_traffic': min_bytes / avg / 1e9,
                          'roofline_bound_us': bound_us, 'frac_of_roofline': bound_us / (avg * 1e6)})
        table.sort(key=lambda r: -r['total_ms'])
        # the roofline kernel is fixed by the WORKLOAD, not by timing noise: the FORWARD edge-stage launch (the metric's
        # "scatter-add ... vs HBM roofline" = the aggregation) that moves the most algorithmic bytes, i.e. the level-0 forward
        # kernel of the mesh (with ~3 bracketed launches per shape in a short run the "largest total time" used before
        # flipped between shapes from run to run); the backward kernels are in edge_kernels
        fwd_rows = [r for r in table if r['kernel'].startswith('stin_edge_relu_mean_fwd')] or table
        dom = max(fwd_rows, key=lambda r: r['algorithmic_MB'])
        edge_total_ms = sum(r['total_ms'] for r in table) / (sum(r['launches'] for r in table) / per_step)
        roofline = {'bound': 'hbm', 'kernel': '%s[N=%d,E=%d,H=%d]' % (dom['kernel'], dom['N'], dom['E'], dom['H']),
                    'achieved': dom['GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': dom['GBps'] / HBM_PEAK_GBS,
                    'traffic': None,
                    'traffic_unit': 'not measured in this run (the live rocprofv3 --pmc passes run only in the default single-GPU fp32 '
                                    'headline run; committed passes of every edge kernel: profiles/r*_pmc_traffic.json = %s bytes for this kernel)'
                                    % pmc_traffic_bytes(dom['kernel'], dom['N'], dom['E'], dom['H']),
                    'avg_us': dom['avg_us'], 'each_us': dom['each_us'], 'algorithmic_bytes': dom['algorithmic_MB'] * 1e6,
                    'convention': 'algorithmic bytes (SURVEY 8d): every gathered row charged once per edge; at 200k vertices the '
                                  'gathered operand (102 MB) is Infinity-Cache resident - see hbm_honest for the HBM-served size',
                    'selection': 'the forward edge-stage (aggregation, HBM-bound) launch with the most algorithmic bytes = the '
                                 'level-0 forward kernel (chosen by the workload, not by measured time); every edge kernel/shape '
                                 'incl. the backward ones is in edge_kernels, the GEMMs (MFMA side, a larger share of the step) '
                                 'are in roofline_gemm'}
        contention = None
        ck = [(k, v) for k, v in ctimes.items() if k[0] == dom['kernel'] and k[1] == (dom['N'], dom['E'], dom['H'])]
        if ck:
            cavg = sum(ck[0][1]) / len(ck[0][1])
            contention = {'kernel': roofline['kernel'], 'avg_us': cavg * 1e6, 'frac': dom['algorithmic_MB'] * 1e6 / cavg / 1e9 / HBM_PEAK_GBS,
                          'slowdown_vs_roofline_bracket': cavg * 1e6 / dom['avg_us'],
                          'note': 'the roofline kernel bracketed in a step whose NEXT plan is being built on the side streams at the '
                                  'same time (prefetch on, as in the un-bracketed timed steps); the `roofline` bracket runs with the '
                                  'side streams idle'}
        out = {
            'metric': 'vertices/sec forward+backward on 200k-vert ScanNet mesh; scatter-add GB/s vs HBM roofline',
            'value': total_vertices * args.steps / dt, 'unit': 'vertices/s', 'n_gpus': ranks_counted, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'SurfaceTextureInpaintingNet 3-D config (ngf 64, n_levels %d, n_blocks 9, ' % cfg['n_levels'] +
                                   'edgeconvtransinv, instance norm, max pool, dilations 1-16), synthetic %d-vertex / '
                                   '%d-directed-edge %d-level mesh per GPU, %s; step = CSR plan build + fwd + '
                                   'masked L1 + bwd + grad all-reduce + Adam(amsgrad)'
                                   % (n0, e0, args.levels, 'fp32' if args.dtype == 'f32' else 'bf16 activation storage '
                                      '(fp32 accumulate / statistics / weights)'),
                       'vertices_per_gpu': n0, 'edges_per_gpu': e0, 'levels': args.levels, 'params': sum(p.numel() for p in net.parameters()),
                       'parallelism': 'dp%d' % world, 'plan_build_in_step': not args.cache_plan,
                       'plan_prefetched_one_step_ahead': not (args.cache_plan or args.no_prefetch_plan or args.graph),
                       'hip_graph': bool(args.graph),
                       'crops_per_step': args.crops or None, 'irregular_mesh': bool(args.irregular)},
            'gemm_precision': ({'fwd': SF.PREC_NAMES[SF.PREC_FWD], 'bwd': SF.PREC_NAMES[SF.PREC_BWD],
                                'note': 'fp32 storage, operands split into 16-bit pieces on the MFMA path (fp16x3: 22-bit '
                                        'products; bf16x3: 16-bit products), fp32 accumulate'} if args.dtype == 'f32' else
                               {'fwd': 'bf16', 'bwd': 'bf16', 'note': 'bf16 storage, one bf16 MFMA per k-step, fp32 accumulate'}),
            'distributed': {'world_size': world, 'backend': (dist.get_backend() if world > 1 else None),
                            'ranks_counted_by_allreduce': ranks_counted, 'ms_per_step_per_rank': rank_ms,
                            'allreduce_us': allreduce_us, 'replicas_bit_identical': identical, 'cores_per_rank': cores_per_rank,
                            'rccl': (rccl_debug_parse(rccl_log, step.bucket.flat.numel() * 4) if rccl_log else None),
                            'allreduce_overlap_validated_on_hardware': False},
            # one row of north_star's 1/2/4/8 table: absolute throughput and the aggregation's share of N x the one-GPU HBM roofline
            # (the driver computes scaling efficiency itself from the per-N `value`s)
            # synchronous data parallelism runs at the pace of the slowest rank: per-rank step time over its mean (1.0 = balanced)
            'straggler': {'vertices_per_rank': 'equal' if not (args.unequal_scenes and world > 1) else
                          [150_000 + (50_000 * r) // (world - 1) for r in range(world)],
                          'allreduce_bracket_us_per_rank': ar_wait_us,
                          'note': 'a synchronous step costs what its largest scene costs (the per-rank wall times equalise); the ranks with '
                                  'smaller scenes spend the difference waiting inside the gradient all-reduce - allreduce_bracket_us_per_rank '
                                  '(HIP events around the end-of-backward all-reduce, mean over the timed steps, one entry per rank).  '
                                  'loader.shard_indices(sizes=...) gives the ranks of one step neighbours in size: profiles/r05_straggler.json'},
            'scaling_table_row': {'gpus': ranks_counted, 'vertices_per_s': total_vertices * args.steps / dt,
                                  'vertices_per_s_per_gpu': total_vertices * args.steps / dt / max(ranks_counted, 1),
                                  'scatter_add_GBps_sum_over_gpus': sum_gbps, 'hbm_roofline_GBps': HBM_PEAK_GBS * world,
                                  'frac_of_n_gpu_hbm_roofline': sum_gbps / (HBM_PEAK_GBS * world),
                                  'kernel': 'level-0 forward edge stage (algorithmic bytes, SURVEY 8d), HIP-event bracket on every rank'},
            'dtype_tolerance': (
                {'forward_max_abs_vs_cpu_oracle': 1e-4, 'loss_abs': 1e-6, 'weight_grad_rel_l2': 1e-3,
                 'measured_at_this_size': 'fwd 6.5e-6, grad rel-L2 3.0e-4 (tests/test_full_size_parity.py, two seeds)'}
                if args.dtype == 'f32' else
                {'forward_max_abs_vs_fp32_oracle': 0.115, 'forward_mean_abs': 1.5e-2, 'loss_rel': 5e-4, 'weight_grad_rel_l2': 0.24,
                 'measured': 'fwd max-abs 7.6e-2, mean-abs 9.9e-3, loss 2.1e-4, grad rel-L2 16.1 % on the 15-block network (bars = 1.5 x: '
                             'tests/test_hip_bf16.py::test_bf16_network_vs_fp32_oracle_stated_tolerance); five-level network (config 5) vs the '
                             'fp32 oracle at 30 k vertices: 1.49e-1 / 1.78e-2 / 1.2e-4 / 27.4 % (tests/test_five_level.py, bars 1.5 x); 1 M '
                             'vertices / 5 levels vs the fp32-storage run: max-abs 0.20 (bar 0.31), mean-abs 1.7e-2 (bar 2.6e-2)',
                 'training_curve_200_steps': {'statement': 'mean loss of the last 50 of 200 Adam steps within `last_50_steps` of the NEARER of two '
                                                           'fp32-storage runs (shipped split GEMMs / exact-fp32 GEMMs), which themselves end up to '
                                                           '`fp32_orders_apart` apart: the trajectories are chaotic',
                                              'single_scene': {'first_50_steps': 1e-2, 'last_50_steps': 5e-2, 'fp32_orders_apart': 8e-2,
                                                               'measured': 'bf16 +1.1 .. +3.3 %, fp32 orders -4.3 .. +2.6 % (3 seeds, 2 code states)'},
                                              'crop_batches_4_levels': {'first_50_steps': 2e-2, 'last_50_steps': 8e-2, 'fp32_orders_apart': 12e-2,
                                                                        'measured': 'bf16 -6.4 .. +2.7 %, fp32 orders -8.1 .. +0.7 %'},
                                              'test': 'tests/test_hip_bf16.py::test_bf16_training_curve_tracks_fp32 (3 seeds each)'},
                 'note': 'bf16 ACTIVATION STORAGE (fp32 accumulate, statistics, master weights): a stated-tolerance mode of configs 3 / 5, '
                         'never the headline; one forward rounding per block already gives ~10 % gradient rel-L2 on this network '
                         '(profiles/r02_bf16_sensitivity.md) - what certifies the mode is the training curve'}),
            'loss': float(loss),
            'priming_steps': priming,
            'brackets_inside_network_call': bool(in_net),
            'host_enqueue_ms_per_step': dt_enqueue / args.steps * 1e3,   # < ms_per_step: the GPU, not the host, bounds the step
            'host_thread_cpu_ms_per_step': dt_cpu / args.steps * 1e3,    # CPU time of the main thread only (backward's Python runs in autograd's device thread)
            'fwd_loss_bwd_only': None if args.no_secondary else {
                'ms_per_step': dt_fb / args.steps * 1e3, 'vertices_per_s_per_gpu': n0 * args.steps / dt_fb,
                'note': 'same scene, CSR plan reused, no gradient all-reduce, no optimizer step (rank 0)'},
            'inference': None,
            'roofline': roofline,
            'contention': contention,
            'edge_stage_ms_per_step': edge_total_ms,
            'edge_kernels': table[:6],
        }
        if gemms:
            gemms.sort(key=lambda r: -r['total_ms'])
            flops = sum(2.0 * r['M'] * r['Nc'] * r['K'] * r['launches'] for r in gemms)
            tsum = sum(r['total_ms'] for r in gemms) * 1e-3
            mult = 1 if args.dtype == 'bf16' else 3
            g0 = gemms[0]
            out['roofline_gemm'] = {
                'bound': 'mfma', 'kernel': '%s[M=%d,Nc=%d,K=%d]' % (g0['kernel'], g0['M'], g0['Nc'], g0['K']),
                'achieved': g0['mfma_TFLOPs_executed'], 'peak': MFMA_16BIT_PEAK_TF, 'unit': 'TFLOP/s',
                'frac': g0['mfma_TFLOPs_executed'] / MFMA_16BIT_PEAK_TF, 'avg_us': g0['avg_us'],
                'roofline_bound_us': g0['roofline_bound_us'], 'frac_of_roofline': g0['frac_of_roofline'],
                'note': 'the GEMM shape with the largest total time; achieved = executed 16-bit MFMA flops (%d per fp32 product) '
                        '/ stand-alone duration; roofline_bound_us = max(min HBM bytes / %.2f TB/s copy rate, executed flops / '
                        '2.5 PF) - these tall-skinny shapes are bounded by their output bytes' % (mult, HBM_COPY_GBS / 1e3),
                'mfma_busy': 'SQ-counter MFMA-busy per kernel is a separate rocprofv3 --pmc pass: profiles/r*_pmc_mfma.md (not part of this run)'}
            out['gemm'] = {'ms_per_step': tsum / gemm_steps * 1e3, 'GFLOP_per_step': flops / gemm_steps / 1e9,
                           'TFLOPs_fp32_equivalent': flops / tsum / 1e12, 'mfma_TFLOPs_executed': mult * flops / tsum / 1e12,
                           'frac_of_16bit_mfma_peak': mult * flops / tsum / 1e12 / MFMA_16BIT_PEAK_TF,
                           'frac_of_f32_mfma_peak': flops / tsum / 1e12 / MFMA_F32_PEAK_TF,
                           'time_weighted_frac_of_roofline': sum(r['roofline_bound_us'] * r['launches'] for r in gemms) /
                                                             sum(r['avg_us'] * r['launches'] for r in gemms),
                           'kernels': gemms[:24]}
        live = None
        if world == 1 and not args.no_secondary and args.dtype == 'f32' and (n0, e0) == (200704, 1200642) and not args.no_live_traffic:
            live_all, why = measure_fabric_traffic()
            live = live_all['headline'] if live_all is not None else None
            if live is not None:
                out['roofline']['traffic'] = live
                out['roofline']['traffic_unit'] = ('FABRIC bytes per launch incl. Infinity-Cache hits ((2*FETCH_SIZE + WRITE_SIZE) KiB), MEASURED in '
                                                   'this run: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) over the '
                                                   'same kernel at the same shape (profiles/pmc_kernels.py, PMC_LIVE=1), mean of the last 3 launches')
                out['roofline']['traffic_over_algorithmic'] = live / (out['roofline']['algorithmic_bytes'] + e0 * 128 / 8.0)
            else:
                out['roofline']['traffic_live_pass'] = 'not available (%s)' % why
        if world == 1 and not args.no_secondary:
            out['scatter_add'] = scatter_add_standalone(device)
            out['hbm_honest'] = hbm_honest_edge_kernel(device, live_bytes=(live_all or {}).get('n1m') if live is not None else None)
            if not (args.graph or args.cache_plan):
                # GPU idle per step WITHOUT a profiler: the wall step minus the same step with the host out of the way
                gpu_ms, host_ms = backlogged_step_ms(one_step)
                out['gpu_idle'] = {'ms_per_step_wall': dt / args.steps * 1e3, 'ms_per_step_gpu_backlogged': gpu_ms,
                                   'idle_ms_per_step': max(0.0, dt / args.steps * 1e3 - gpu_ms), 'host_enqueue_ms_per_step_backlogged': host_ms,
                                   'method': 'a ~60 ms sleep kernel holds the compute stream while the host enqueues 10 steps behind it; HIP '
                                             'events on that stream around the 10 steps = GPU time per step with no launch-side bubbles '
                                             '(side-stream work included, as in the timed steps); idle = ms_per_step - that figure '
                                             '(un-profiled; rocprofv3 traces in profiles/ slow the host and show more)'}
            if args.dtype == 'f32' and not (args.no_exact_fp32 or args.crops or args.graph or args.irregular or args.morton_order or
                                             args.coherent_order or os.environ.get('STIN_GEMM_FWD') == '0'):
                out['exact_fp32'] = exact_fp32_companion(args)
            try:                                            # scipy (Delaunay) is an optional dependency of this one leg
                out['roofline_irregular'] = irregular_edge_kernel(device)
            except Exception as exc:                        # noqa: BLE001 - the JSON line must survive a missing optional package
                out['roofline_irregular'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
            if world == 1 and not (args.crops or args.morton_order or args.coherent_order or args.graph):
                # what vertex LOCALITY is worth: the same scene renumbered once on the host (synthetic.renumber_by_locality: level 0
                # by the Morton code of its positions, coarser levels by their first child - what a reader can do at load time),
                # a few plain steps.  The benchmark meshes are randomly numbered on purpose (ScanNet-like order), the
                # headline value stays on that order.
                from surface_texture_inpainting_net_amd.synthetic import renumber_by_locality
                loc = renumber_by_locality(sample.to('cpu'))[0].to(device)
                for _ in range(3):
                    loc._plan_cache = None
                    step(loc)
                fence()
                t2 = time.perf_counter()
                for _ in range(10):
                    loc._plan_cache = None
                    step(loc)
                fence()
                out['vertex_locality'] = {'ms_per_step': (time.perf_counter() - t2) / 10 * 1e3,
                                          'note': 'same scene, vertices of every level renumbered by locality on the host before the run '
                                                  '(Morton order of the positions; not part of the timed step, not the headline)'}
                try:
                    out['loader_fed'] = loader_fed_step(device, net, step, args.vertices, args.levels)
                except Exception as exc:                    # noqa: BLE001 - a secondary leg must not lose the line
                    out['loader_fed'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
            # forward only, as the reference's validation loop runs it (trainers/inpainting3d_trainer.py:204-263: model.eval(),
            # model(data) under torch.no_grad()): CSR plan resident, no ReLU mask stored, block temporaries shared (functional.NetFn,
            # need_grad False).  Runs AFTER every training leg: its differently sized arenas re-cut the caching allocator's blocks
            # (a training leg measured right behind it paid hipMalloc / hipFree in every step: 20 ms instead of 7).
            net.eval()
            with torch.no_grad():
                for _ in range(3):
                    net(sample)
                fence()
                t_inf = time.perf_counter()
                for _ in range(args.steps):
                    net(sample)
                fence()
                dt_inf = time.perf_counter() - t_inf
            net.train()
            torch.cuda.empty_cache()
            out['inference'] = {'ms': dt_inf / args.steps * 1e3, 'vertices_per_s': n0 * args.steps / dt_inf,
                                'note': 'eval mode, torch.no_grad(), plan resident, rank 0; no ReLU mask store, shared block temporaries'}
            if not args.no_cpu_baseline:
                out['cpu_baseline'] = cpu_baseline(args.vertices, args.levels, seed=0, headline_mesh=not args.quick_cpu_baseline)
        print(compact_line(out, write_detail(out, args.detail)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if not identical:
            sys.exit(3)


if __name__ == '__main__':
    main()
