#!/bin/bash
# rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE - separate passes, --kernel-trace only) over the SingleConvMeshNet step:
# fabric bytes per launch of its kernels -> gpurun_out/<tag>/pmc_scmn.json (the gather / segment kernels are what r06_scmn.md quotes)
#   gpurun --timeout 900 -- 'bash profiles/pmc_scmn.sh r06'
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_scmn_fetch -o run -- python3 $R/profiles/scmn_bench.py --steps 3 --warmup 2 > $O/pmc_scmn_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_scmn_write -o run -- python3 $R/profiles/scmn_bench.py --steps 3 --warmup 2 > $O/pmc_scmn_write.log 2>&1
F=$(find $O/pmc_scmn_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_scmn_write -name '*counter_collection.csv' | head -1)
python3 $R/profiles/pmc_summarize.py $F $W > $O/pmc_scmn.json
rm -rf $O/pmc_scmn_fetch $O/pmc_scmn_write
python3 -c "
import json
t = json.load(open('$O/pmc_scmn.json'))
for k, v in sorted(t.items(), key=lambda kv: -kv[1]['fabric_MB_per_launch'])[:14]:
    print('%-70s fabric %8.1f MB (read %8.1f, write %8.1f)' % (k[:70], v['fabric_MB_per_launch'], v['fabric_read_MB_corrected'], v['fabric_write_MB']))"
