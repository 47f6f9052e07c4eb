#!/bin/bash
# Copy one collect_round.sh result directory (gpurun_out/<dir>) into the tracked profiles/<round>_* files.
#   bash profiles/publish_round.sh r03 [r03]        (source directory under gpurun_out/, round prefix)
set -e
S=gpurun_out/${1:-r06}
RN=${2:-r06}
P=profiles/$RN
cp $S/bench_default.json ${P}_bench_default.json; cp $S/bench_final.json ${P}_bench_final.json; cp $S/bench_bf16.json ${P}_bench_bf16.json
cp $S/bench_irregular.json ${P}_bench_irregular.json
cp $S/bench_default_line.json ${P}_bench_default_line.json; cp $S/bench_final_line.json ${P}_bench_final_line.json   # the compact stdout lines (what the driver parses)
for c in c2 c3 c5_bf16 c5_f32; do cp $S/config_$c.json ${P}_config_$c.json; done
cp $S/small_20k_eager.json ${P}_small_20k_eager.json; cp $S/small_20k_graph.json ${P}_small_20k_graph.json
cp $S/gemm_shapes.md ${P}_gemm_shapes.md; cp $S/tn_ws.md ${P}_tn_ws.md
cp $S/prof/run_kernel_stats.csv ${P}_bench_final_kernel_stats.csv
for c in c2 c3 c5; do cp $S/prof_$c/run_kernel_stats.csv ${P}_config_${c}_kernel_stats.csv; done
python3 - "$S" "$RN" <<'PY'
import csv, json, subprocess, sys
S, RN = sys.argv[1], sys.argv[2]
P = 'profiles/' + RN


def line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def tool(name, *args):
    return subprocess.run([sys.executable, 'profiles/' + name] + list(args), capture_output=True, text=True).stdout


def level0_fwd_avg_us(stats_csv):
    """Average duration of the level-0 forward edge kernel in a rocprofv3 kernel_stats.csv: of the k_edge_fwd* rows the one
    with the largest average (level 0 has the most edges x channels of any level)."""
    best = None
    for r in csv.DictReader(open(stats_csv)):
        if 'k_edge_fwd' in r['Name']:
            avg = int(r['TotalDurationNs']) / int(r['Calls'])
            if best is None or avg > best[0]:
                best = (avg, r['Name'], int(r['Calls']))
    return best[0] / 1e3, best[2]


d, dd = line(S + '/bench_final.json'), line(S + '/bench_default.json')
head = ('# Round %s, fp32 headline: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 5 '
        '--no-cpu-baseline --no-secondary` (200 704-vertex mesh, 1x MI355X; un-profiled `python bench.py --steps 30 --warmup 5`: '
        '%.2f ms/step = %.1f M vertices/s, %s_bench_final.json; the no-flag default run: %s_bench_default.json, %.2f ms).\n\n'
        'Under the profiler the host is slower than un-profiled, so the timeline below carries more idle time than the bench line; '
        'kernel durations are what the table is for.  The weight-gradient work (`k_gemm_tn_ws`, `k_gemm_tn_*`, `k_wgrad_finalize`, '
        '`k_reduce_slabs`) and the next step\'s plan build (`k_count`, `k_rows`, `k_fill`, `k_rank`) run on side streams beside '
        'the critical-path kernels: durations SUM to more than the step, and co-running kernels are individually slower than alone '
        '(`k_gemm_tn_ws` 18063x1024x256: ~46 us alone, ~90 us beside the dx chain; `k_colreduce<float, 2, 4>` ~9 us alone, 23 us in '
        'the step) - stand-alone figures are in %s_gemm_shapes.md / %s_tn_ws.md.\n\n```\n'
        % (RN[1:], d['ms_per_step'], d['value'] / 1e6, RN, RN, dd['ms_per_step'], RN, RN))
gaps = tool('gaps.py', S + '/prof/run_kernel_trace.csv', '--steps', '8')
summ = tool('summarize.py', S + '/prof/run_kernel_stats.csv', '16')
open(P + '_bench_final.md', 'w').write(head + '\n'.join(gaps.splitlines()[:8]) + '\n```\n\n' + summ)

# kernel tables of configs 2 / 3 / 5 and the roofline cross-check (bench line's HIP-event bracket vs rocprofv3's average)
cfg = [('c2', 'config_c2', 'prof_c2', '2: one ~150 k-vertex scene, 3 levels, fp32', '--vertices 150000', 14),
       ('c3', 'config_c3', 'prof_c3', '3: 8 unequal crops, 4 levels, bf16', '--crops 8 --levels 4 --dtype bf16', 14),
       ('c5', 'config_c5_bf16', 'prof_c5', '5: 1 M vertices, 5 levels, bf16', '--vertices 1000000 --levels 5 --dtype bf16', 11)]
check = ['| config | kernel (bench line) | algorithmic MB / launch | bracket avg us (un-profiled run) | `roofline.frac` (JSON) | '
         'rocprofv3 avg us (calls) | frac from rocprofv3 | JSON / rocprof |', '|---|---|---|---|---|---|---|---|']
for tag, jf, pd, title, flags, nsteps in [('hl', 'bench_final', 'prof', 'headline: 200 704 vertices, 3 levels, fp32', '', 16)] + cfg:
    r = line('%s/%s.json' % (S, jf))['roofline']
    us, calls = level0_fwd_avg_us('%s/%s/run_kernel_stats.csv' % (S, pd))
    f_prof = r['algorithmic_bytes'] / (us * 1e-6) / 1e9 / r['peak']
    check.append('| %s | `%s` | %.1f | %.1f | %.3f | %.1f (%d) | %.3f | %.3f |' % (
        title, r['kernel'], r['algorithmic_bytes'] / 1e6, r['avg_us'], r['frac'], us, calls, f_prof, r['frac'] / f_prof))
    if tag != 'hl':
        body = ('# Round %s, config %s under `rocprofv3 --kernel-trace --stats`: `bench.py %s --no-cpu-baseline --no-secondary` '
                '(%d profiled steps incl. warm-up)\n\n```\n' % (RN[1:], title, flags, nsteps))
        body += '\n'.join(tool('gaps.py', '%s/%s/run_kernel_trace.csv' % (S, pd), '--steps', str(nsteps - 4)).splitlines()[:6]) + '\n```\n\n'
        body += tool('summarize.py', '%s/%s/run_kernel_stats.csv' % (S, pd), str(nsteps))
        open('%s_config_%s_kernel_stats.md' % (P, tag), 'w').write(body)

rows = [('2: one ~150 k-vertex scene, 3 levels, fp32', '`bench.py --vertices 150000 --steps 30 --warmup 5`', 'config_c2'),
        ('3: batch of 8 unequal crops (161 k vertices), 4 levels, bf16', '`bench.py --crops 8 --levels 4 --dtype bf16 --steps 30 --warmup 5`', 'config_c3'),
        ('5: 1 M vertices / 6 M edges, 5 levels, bf16 (one rank of the 8)', '`bench.py --vertices 1000000 --levels 5 --dtype bf16 --steps 8 --warmup 3`', 'config_c5_bf16'),
        ('5 in fp32', 'same without `--dtype bf16`', 'config_c5_f32'),
        ('headline in bf16 storage', '`bench.py --steps 30 --warmup 5 --dtype bf16`', 'bench_bf16'),
        ('headline shape on the irregular (Delaunay, degree 3..~20) hierarchy', '`bench.py --steps 30 --warmup 5 --irregular`', 'bench_irregular'),
        ('20 k-vertex crop, eager', '`bench.py --vertices 20000 --steps 40 --warmup 5`', 'small_20k_eager'),
        ('20 k-vertex crop, HIP graph', '`bench.py --vertices 20000 --steps 40 --warmup 5 --graph`', 'small_20k_graph')]
out = ['# Round %s: the other BASELINE configurations on 1 x MI355X (parity-test cases, not the bench line)' % RN[1:], '',
       'All with `--no-cpu-baseline --no-secondary`; one box, back to back (`profiles/collect_round.sh`, published with '
       '`profiles/publish_round.sh`).  Host enqueue time per step is box-dependent (+-0.5 ms for the same step on different boxes of '
       'the pool).', '',
       '| config | command | ms/step | vertices/s | host enqueue ms | `roofline.frac` (level-0 forward, bracketed) |', '|---|---|---|---|---|---|']
for name, cmd, f in rows:
    r = line('%s/%s.json' % (S, f))
    out.append('| %s | %s | %.2f | %.1f M | %.2f | %.3f |' % (name, cmd, r['ms_per_step'], r['value'] / 1e6, r['host_enqueue_ms_per_step'],
                                                            r['roofline']['frac']))
out += ['', '## `roofline.frac`: the bench line\'s bracket against rocprofv3', '',
        'The bench line brackets ONE step without plan prefetch (so the level-0 forward kernel is timed alone on its stream, not beside '
        'the next step\'s plan build - the round-2 artefact that read 0.26 at config 5); the rocprofv3 column is the average of the same '
        'kernel over every profiled step of a separate run (`%s_config_c{2,3,5}_kernel_stats.csv` / `.md`, `%s_bench_final_kernel_stats.csv`), '
        'where prefetch is on: its level-0 launches can overlap the plan build of the next step.' % (RN, RN), ''] + check
out += ['', 'Round 4 for comparison (r04_configs.md): config 2 5.9 ms, config 3 5.46 ms, config 5 23.6 ms (bf16) / 41.1 ms (fp32), headline 7.13-7.3 ms, bf16 4.79 ms, 20 k-vertex crop 2.17 ms as a HIP graph.',
        '', 'JSON lines: `%s_config_{c2,c3,c5_bf16,c5_f32}.json`, `%s_bench_{bf16,irregular}.json`, `%s_small_20k_{eager,graph}.json`.' % (RN, RN, RN)]
open(P + '_configs.md', 'w').write('\n'.join(out) + '\n')
# SingleConvMeshNet: the bench line and the kernel table of the same command
import os
if os.path.exists(S + '/scmn.json') and os.path.exists(S + '/prof_scmn/run_kernel_stats.csv'):
    sc = line(S + '/scmn.json')
    open(P + '_scmn.json', 'w').write(json.dumps(sc) + '\n')
    open(P + '_scmn_kernel_stats.csv', 'w').write(open(S + '/prof_scmn/run_kernel_stats.csv').read())
    head = ('# Round %s: SingleConvMeshNet (SURVEY 8f rank 3) training step, %d vertices, filters 64/128/256, 2 propagation steps, fp32, '
            'Adam: `python profiles/scmn_bench.py` = %.2f ms per step = %.1f M vertices/s, %.2f GB peak (round 4: 21.8 ms, 11.74 GB; round 1: 22.6 ms; '
            '`%s_scmn.json` incl. the level-0 rooflines); kernel table of the same command under `rocprofv3 --kernel-trace --stats` (14 steps incl. '
            'warm-up and the bracketed one)\n\n' % (
                RN[1:], sc['vertices'], sc['ms_per_step'], sc['vertices_per_s'] / 1e6, sc['peak_gb'], RN))
    open(P + '_scmn.md', 'w').write(head + tool('summarize.py', S + '/prof_scmn/run_kernel_stats.csv', '14'))
print(d['ms_per_step'], dd['ms_per_step'])
PY
