#!/bin/bash
# SQ counters of the NT kernels at the bottleneck shapes, two rocprofv3 --pmc passes per shape (8 SQ slots each) over
# profiles/pmc_nt_wide.py -> gpurun_out/<tag>/pmc_nt_panel.txt      gpurun --timeout 900 -- 'bash profiles/pmc_nt_panel.sh r04'
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_nt_panel.txt
for SH in ${SHAPES:-18063,256,1024 18063,512,256}; do
  export SHAPE=$SH
  i=0
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    rm -rf $O/pmc_ntp_${SH}_$i
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc_ntp_${SH}_$i -o run -- python3 $R/profiles/pmc_nt_wide.py > /dev/null 2>&1
    echo "## shape M,Nc,K = $SH pass $i" >> $O/pmc_nt_panel.txt
    python3 $R/profiles/pmc_any_summarize.py $(find $O/pmc_ntp_${SH}_$i -name '*counter_collection.csv') >> $O/pmc_nt_panel.txt
    rm -rf $O/pmc_ntp_${SH}_$i
  done
done
cat $O/pmc_nt_panel.txt
