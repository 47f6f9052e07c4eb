#!/bin/bash
# Fabric traffic (two rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE) of SingleConvMeshNet's per-edge backward at the level-0 shape:
# the three-launch route against the two passes of the streaming-rows kernel (profiles/probes/bnbwd_probe.py) -> gpurun_out/<tag>/pmc_stream.json
#   gpurun --timeout 900 -- 'bash profiles/pmc_stream.sh r05'
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PROBE_OCC=2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_sf -o run -- python3 $R/profiles/probes/bnbwd_probe.py 1200642 128 64 > $O/pmc_stream_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_sw -o run -- python3 $R/profiles/probes/bnbwd_probe.py 1200642 128 64 > $O/pmc_stream_write.log 2>&1
F=$(find $O/pmc_sf -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_sw -name '*counter_collection.csv' | head -1)
python3 $R/profiles/pmc_summarize.py $F $W > $O/pmc_stream.json
rm -rf $O/pmc_sf $O/pmc_sw
python3 -c "
import json
t = json.load(open('$O/pmc_stream.json'))
for k, v in sorted(t.items()):
    print('%-60s %8.1f MB  (read %.1f, write %.1f)' % (k[:60], v['fabric_MB_per_launch'], v['fabric_read_MB_corrected'], v['fabric_write_MB']))"
