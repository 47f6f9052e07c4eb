cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b/pmc; mkdir -p $O
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/p1 -o run -- python3 $R/profiles/pmc_tn.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p2 -o run -- python3 $R/profiles/pmc_tn.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_REQ_sum --kernel-trace --output-format csv -d $O/p3 -o run -- python3 $R/profiles/pmc_tn.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p4 -o run -- python3 $R/profiles/pmc_tn.py > /dev/null 2>&1
python3 $R/profiles/pmc_any_summarize.py $(find $O -name "*counter_collection.csv")
