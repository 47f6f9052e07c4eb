# same-box A/B of the headline step: bash profiles/probes/ab_bench.sh "ENV1=.. ENV2=.." "ENV3=.." ...   ("-" = defaults); two rounds, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    ms=$(env $e python3 $R/bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  host %.2f  loss %.9f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss']))")
    echo "round $round [$v]: $ms"
  done
done
