cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e/prof_c3; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/bench.py --crops 8 --levels 4 --dtype bf16 --steps 10 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/err.log
python3 $R/profiles/gaps.py $O/run_kernel_trace.csv --steps 8 | head -10
python3 $R/profiles/summarize.py $O/run_kernel_stats.csv 16 | head -50
