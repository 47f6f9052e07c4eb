# kernel-trace profile of the headline step: bash profiles/prof_step.sh <tag> [extra bench.py args]; env is inherited (A/B switches)
TAG=${1:-r04}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG/prof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary "$@" > $O/bench.json 2> $O/err.log
python3 $R/profiles/gaps.py $O/run_kernel_trace.csv --steps 8 --context ${GAP_CONTEXT_US:-0} | head -${GAP_LINES:-14}
python3 $R/profiles/summarize.py $O/run_kernel_stats.csv 16 | head -${LINES_OUT:-70}
rm -f $O/run_kernel_trace.csv
