#!/bin/bash
# MFMA-busy share of the NT kernels at the bottleneck shapes: one rocprofv3 --pmc pass per shape (SQ_VALU_MFMA_BUSY_CYCLES counts
# cycles, SQ_BUSY_CU_CYCLES cycles x CUs) -> gpurun_out/<tag>/pmc_mfma.txt       gpurun --timeout 600 -- 'bash profiles/pmc_mfma_busy.sh r03'
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_mfma.txt
for SH in 18063,256,1024 18063,1024,256 18063,512,256 18063,256,512; do
  export SHAPE=$SH
  rm -rf $O/pmc_mfma_$SH
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_mfma_$SH -o run -- python3 $R/profiles/pmc_nt_wide.py > /dev/null 2>&1
  echo "## shape M,Nc,K = $SH" >> $O/pmc_mfma.txt
  python3 $R/profiles/pmc_any_summarize.py $(find $O/pmc_mfma_$SH -name '*counter_collection.csv') >> $O/pmc_mfma.txt
done
cat $O/pmc_mfma.txt
