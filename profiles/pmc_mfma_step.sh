#!/bin/bash
# MFMA-busy share of every GEMM kernel of the headline step (round 6: k_gemm_tn_ws, the narrow-tile TN products, the panel / strip /
# stream NT kernels) and of the SingleConvMeshNet step: one rocprofv3 --pmc pass each (SQ_VALU_MFMA_BUSY_CYCLES counts cycles,
# SQ_BUSY_CU_CYCLES cycles x CUs; --kernel-trace only)  ->  gpurun_out/<tag>/pmc_mfma_step.md
#   gpurun --timeout 900 -- 'bash profiles/pmc_mfma_step.sh r06'
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_mfma_hl $O/pmc_mfma_scmn
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_mfma_hl -o run -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-secondary --detail $O/pmc_mfma_hl.json > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_mfma_scmn -o run -- python3 $R/profiles/scmn_bench.py --steps 3 --warmup 2 > /dev/null 2>&1
python3 - "$O" <<'PY'
import collections, csv, glob, re, sys
O = sys.argv[1]
out = ['# MFMA-busy share per GEMM kernel inside the step (round 6; `bash profiles/pmc_mfma_step.sh`: one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES '
       'SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace` pass over the whole step, every launch of the profiled steps)', '',
       'MFMA busy (chip) = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.1 GHz x 1024 SIMDs); "of the CUs in use" divides by SQ_BUSY_CU_CYCLES x 4 SIMDs.  '
       'Durations are in-step (co-running side-stream kernels included) and under the counter pass.', '']
for tag, title in (('pmc_mfma_hl', 'headline step (200 704 vertices, fp32 storage, fp16x3 / bf16x3 products)'), ('pmc_mfma_scmn', 'SingleConvMeshNet step (200 704 vertices)')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(O + '/' + tag + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            name = re.sub(r'\(.*', '', re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', r['Kernel_Name'])))
            m = re.match(r'_ZN12_GLOBAL__N_1\d+(k_\w+?)I(.*?)EEv', name)
            if m:
                name = m.group(1) + '<' + m.group(2)[:24] + '>'
            if 'gemm' not in name:
                continue
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] == 'SQ_WAVES':
                acc[name]['_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out += ['## ' + title, '', '| kernel | launches | avg us | MFMA busy (chip) | MFMA busy of the CUs in use | total ms |', '|---|---|---|---|---|---|']
    rows = []
    for name, cs in acc.items():
        n = len(cs['_us'])
        if not n:
            continue
        us = sum(cs['_us']) / n
        mf = sum(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / n
        cu = sum(cs['SQ_BUSY_CU_CYCLES']) / n
        rows.append((us * n, '| `%s` | %d | %.1f | %.0f %% | %.0f %% | %.2f |' % (name, n, us, 100 * mf / (us * 2.1e3 * 1024), 100 * mf / max(cu * 4, 1), us * n / 1e3)))
    out += [r for _, r in sorted(rows, reverse=True)] + ['']
open(O + '/pmc_mfma_step.md', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/pmc_mfma_hl $O/pmc_mfma_scmn
