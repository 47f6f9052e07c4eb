#!/bin/bash
# Copy one collect_round.sh result directory (gpurun_out/<tag>) into the tracked profiles/r02_* files.
#   bash profiles/publish_round.sh r02e
set -e
S=gpurun_out/${1:-r02}
P=profiles/r02
cp $S/bench_default.json ${P}_bench_default.json; cp $S/bench_final.json ${P}_bench_final.json; cp $S/bench_bf16.json ${P}_bench_bf16.json
for c in c2 c3 c5_bf16 c5_f32; do cp $S/config_$c.json ${P}_config_$c.json; done
cp $S/small_20k_eager.json ${P}_small_20k_eager.json; cp $S/small_20k_graph.json ${P}_small_20k_graph.json
cp $S/gemm_shapes.md ${P}_gemm_shapes.md; cp $S/prof/run_kernel_stats.csv ${P}_bench_final_kernel_stats.csv
python3 - "$S" <<'PY'
import json, subprocess, sys
S = sys.argv[1]
d = json.loads(open(S + '/bench_final.json').read().strip().splitlines()[-1])
dd = json.loads(open(S + '/bench_default.json').read().strip().splitlines()[-1])
head = ('# Round 2, fp32 headline: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 '
        '--no-cpu-baseline --no-secondary` (200 704-vertex mesh, 1x MI355X; un-profiled `python bench.py --steps 30 --warmup 5`: '
        '%.2f ms/step = %.1f M vertices/s, r02_bench_final.json; the no-flag default run: r02_bench_default.json, %.2f ms).\n\n'
        'Under the profiler the host needs ~9 ms per step (it is the limit there), so the timeline below is host-paced; kernel durations '
        'are what the table is for.  The weight-gradient GEMMs (`k_gemm_tn_*`, `k_reduce_slabs`, `k_unpack`) and the next step\'s plan '
        'build (`k_count`, `k_fill`, `k_rank`, rocprim scan ...) run on side streams beside the critical-path kernels: durations SUM to '
        'more than the step and co-running kernels are individually slower than alone (TN 128x128: 54-58 us alone; '
        '`k_gemm_nt_wide<__bf16, 4>` 45 us alone; `k_reduce_slabs` 8 us alone).\n\n```\n' % (d['ms_per_step'], d['value'] / 1e6, dd['ms_per_step']))
gaps = subprocess.run([sys.executable, 'profiles/gaps.py', S + '/prof/run_kernel_trace.csv', '--steps', '8'], capture_output=True, text=True).stdout
summ = subprocess.run([sys.executable, 'profiles/summarize.py', S + '/prof/run_kernel_stats.csv', '13'], capture_output=True, text=True).stdout
open('profiles/r02_bench_final.md', 'w').write(head + '\n'.join(gaps.splitlines()[:8]) + '\n```\n\n' + summ)
rows = [('2: one ~150 k-vertex scene, 3 levels, fp32', '`bench.py --vertices 150000 --steps 30 --warmup 5`', 'config_c2'),
        ('3: batch of 8 unequal crops (161 k vertices), 4 levels, bf16', '`bench.py --crops 8 --levels 4 --dtype bf16 --steps 30 --warmup 5`', 'config_c3'),
        ('5: 1 M vertices / 6 M edges, 5 levels, bf16 (one rank of the 8)', '`bench.py --vertices 1000000 --levels 5 --dtype bf16 --steps 8 --warmup 3`', 'config_c5_bf16'),
        ('5 in fp32', 'same without `--dtype bf16`', 'config_c5_f32'),
        ('headline in bf16 storage', '`bench.py --steps 30 --warmup 5 --dtype bf16`', 'bench_bf16'),
        ('20 k-vertex crop, eager', '`bench.py --vertices 20000 --steps 40 --warmup 5`', 'small_20k_eager'),
        ('20 k-vertex crop, HIP graph', '`bench.py --vertices 20000 --steps 40 --warmup 5 --graph`', 'small_20k_graph')]
out = ['# Round 2: the other BASELINE configurations on 1 x MI355X (parity-test cases, not the bench line)', '',
       'All with `--no-cpu-baseline --no-secondary`; one box, back to back (`profiles/collect_round.sh`, published with '
       '`profiles/publish_round.sh`).  Host enqueue time per step is box-dependent (3.7-5.2 ms for the same 20 k-vertex step on '
       'different boxes of the pool; config 3 sits at the host / GPU boundary: 6.5-7.3 ms).', '',
       '| config | command | ms/step | vertices/s | host enqueue ms |', '|---|---|---|---|---|']
for name, cmd, f in rows:
    r = json.loads(open('%s/%s.json' % (S, f)).read().strip().splitlines()[-1])
    out.append('| %s | %s | %.2f | %.1f M | %.2f |' % (name, cmd, r['ms_per_step'], r['value'] / 1e6, r['host_enqueue_ms_per_step']))
out += ['', 'Round 1 for comparison: config 2 7.58 ms, config 3 7.31 ms, config 5 28.96 ms (bf16) / 47.75 ms (fp32), headline bf16 6.48 ms.',
        '', 'JSON lines: `r02_config_{c2,c3,c5_bf16,c5_f32}.json`, `r02_bench_bf16.json`, `r02_small_20k_{eager,graph}.json`.']
open('profiles/r02_configs.md', 'w').write('\n'.join(out) + '\n')
print(d['ms_per_step'], dd['ms_per_step'])
PY
