# kernel-trace profile of the SingleConvMeshNet step: bash profiles/prof_scmn.sh <tag>; env is inherited (A/B switches)
TAG=${1:-r05}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG/prof_scmn; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/profiles/scmn_bench.py "$@" > $O/bench.json 2> $O/err.log
python3 $R/profiles/summarize.py $O/run_kernel_stats.csv 13 | head -${LINES_OUT:-45}
rm -f $O/run_kernel_trace.csv
