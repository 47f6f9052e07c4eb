cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5r; mkdir -p $O
export STIN_REORDER=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/bench.py --vertices 1000000 --levels 5 --dtype bf16 --steps 4 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/err.log
python3 $R/profiles/summarize.py $O/run_kernel_stats.csv 10 | head -30
