#!/bin/bash
# LDS bank conflicts of the bf16 TN kernels (register-transpose vs transposed-read) at one fat shape: one rocprofv3 --pmc pass.
#   gpurun --timeout 600 -- 'bash profiles/pmc_tn_tr.sh r03'
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/_tn_tr_run.py <<PY
import os, sys, torch
sys.path.insert(0, "$R")
from surface_texture_inpainting_net_amd import functional as SF
G = torch.randn(8100, 4096, device="cuda").bfloat16(); X = torch.randn(8100, 1024, device="cuda").bfloat16()
for env in ({"STIN_TN_TR": "0", "STIN_TN_BIG": "0"}, {"STIN_TN_TR": "1", "STIN_TN_BIG": "0"}, {}):
    for k in ("STIN_TN_TR", "STIN_TN_BIG"): os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(4): SF.gemm_tn(G, X, ones_column=True)
torch.cuda.synchronize()
PY
rm -rf $O/pmc_tn_tr
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_tn_tr -o run -- python3 /tmp/_tn_tr_run.py > /dev/null 2>&1
python3 $R/profiles/pmc_any_summarize.py $(find $O/pmc_tn_tr -name '*counter_collection.csv') | tee $O/pmc_tn_tr.txt
