#!/bin/bash
# Two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one TCC pass) over profiles/pmc_kernels.py -> gpurun_out/<tag>/pmc_traffic.json
#   gpurun --timeout 900 -- 'bash profiles/pmc_traffic.sh r03'
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- python3 $R/profiles/pmc_kernels.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- python3 $R/profiles/pmc_kernels.py > $O/pmc_write.log 2>&1
F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
python3 $R/profiles/pmc_summarize.py $F $W > $O/pmc_traffic.json
python3 -c "
import json; t=json.load(open('$O/pmc_traffic.json'))
for k,v in sorted(t.items()):
    if 'edge' in k or 'segment' in k: print(k, round(v['hbm_MB_per_launch'],1))"
