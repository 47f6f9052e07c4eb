#!/bin/bash
# rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one TCC pass) over profiles/pmc_kernels.py at the headline size and
# at the 1 M-vertex size of bench.py's `hbm_honest` -> gpurun_out/<tag>/pmc_traffic.json (keys of the 1 M pass prefixed "N1000000:")
#   gpurun --timeout 1200 -- 'bash profiles/pmc_traffic.sh r04'
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for NV in 200000 1000000; do
  export PMC_VERTICES=$NV
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$NV -o run -- python3 $R/profiles/pmc_kernels.py > $O/pmc_fetch_$NV.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$NV -o run -- python3 $R/profiles/pmc_kernels.py > $O/pmc_write_$NV.log 2>&1
  F=$(find $O/pmc_fetch_$NV -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write_$NV -name '*counter_collection.csv' | head -1)
  if [ $NV = 200000 ]; then python3 $R/profiles/pmc_summarize.py $F $W > $O/pmc_traffic_$NV.json; else python3 $R/profiles/pmc_summarize.py $F $W "N$NV:" > $O/pmc_traffic_$NV.json; fi
  rm -rf $O/pmc_fetch_$NV $O/pmc_write_$NV
done
# (round 6) the headline mesh renumbered by locality (loader.SceneLoader's resident plans): keys prefixed "locality:"
export PMC_VERTICES=200000 PMC_LOCALITY=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_loc -o run -- python3 $R/profiles/pmc_kernels.py > $O/pmc_fetch_loc.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_loc -o run -- python3 $R/profiles/pmc_kernels.py > $O/pmc_write_loc.log 2>&1
F=$(find $O/pmc_fetch_loc -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write_loc -name '*counter_collection.csv' | head -1)
python3 $R/profiles/pmc_summarize.py $F $W "locality:" > $O/pmc_traffic_loc.json
rm -rf $O/pmc_fetch_loc $O/pmc_write_loc
unset PMC_LOCALITY
python3 -c "
import json
t = json.load(open('$O/pmc_traffic_200000.json')); t.update(json.load(open('$O/pmc_traffic_1000000.json'))); t.update(json.load(open('$O/pmc_traffic_loc.json')))
json.dump(t, open('$O/pmc_traffic.json', 'w'), indent=1)
for k, v in sorted(t.items()):
    if 'edge' in k or 'segment' in k: print(k, round(v['fabric_MB_per_launch'], 1), 'read', round(v['fabric_read_MB_corrected'], 1), 'write', round(v['fabric_write_MB'], 1))"
