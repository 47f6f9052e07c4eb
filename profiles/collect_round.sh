#!/bin/bash
# Collect the round's measured evidence on the GPU box into gpurun_out/<tag>/ (published into profiles/ by publish_round.sh).
#   gpurun --timeout 2400 -- 'bash profiles/collect_round.sh r04'
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --detail $O/bench_default.json > $O/bench_default_line.json 2> $O/bench_default.err
python3 $R/bench.py --steps 30 --warmup 5 --detail $O/bench_final.json > $O/bench_final_line.json 2> $O/bench_final.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary --detail $O/bench_prof.json > $O/bench_prof_line.json 2> $O/prof.err
python3 $R/bench.py --steps 30 --warmup 5 --dtype bf16 --no-cpu-baseline --detail $O/bench_bf16.json > $O/bench_bf16_line.json 2> $O/bench_bf16.err
python3 $R/bench.py --steps 30 --warmup 5 --irregular --no-cpu-baseline --no-secondary --detail $O/bench_irregular.json > $O/bench_irregular_line.json 2>> $O/configs.err
python3 $R/bench.py --vertices 150000 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --detail $O/config_c2.json > $O/config_c2_line.json 2>> $O/configs.err
python3 $R/bench.py --crops 8 --levels 4 --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --detail $O/config_c3.json > $O/config_c3_line.json 2>> $O/configs.err
python3 $R/bench.py --vertices 1000000 --levels 5 --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --detail $O/config_c5_bf16.json > $O/config_c5_bf16_line.json 2>> $O/configs.err
python3 $R/bench.py --vertices 1000000 --levels 5 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --detail $O/config_c5_f32.json > $O/config_c5_f32_line.json 2>> $O/configs.err
python3 $R/bench.py --vertices 20000 --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --detail $O/small_20k_eager.json > $O/small_20k_eager_line.json 2>> $O/configs.err
python3 $R/bench.py --vertices 20000 --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --graph --detail $O/small_20k_graph.json > $O/small_20k_graph_line.json 2>> $O/configs.err
# rocprofv3 kernel traces of configs 2 / 3 / 5: the roofline kernel's average there must agree with the bench line's bracket
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o run -- python3 $R/bench.py --vertices 150000 --steps 8 --warmup 5 --no-cpu-baseline --no-secondary --detail $O/prof_c2.json > $O/prof_c2_line.json 2>> $O/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o run -- python3 $R/bench.py --crops 8 --levels 4 --dtype bf16 --steps 8 --warmup 5 --no-cpu-baseline --no-secondary --detail $O/prof_c3.json > $O/prof_c3_line.json 2>> $O/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o run -- python3 $R/bench.py --vertices 1000000 --levels 5 --dtype bf16 --steps 5 --warmup 5 --no-cpu-baseline --no-secondary --detail $O/prof_c5.json > $O/prof_c5_line.json 2>> $O/prof.err
python3 $R/profiles/gemm_shapes.py --rounds 7 --variants FRAG=1+STIN_NT_PANEL=0,FRAG=1 --md $O/gemm_shapes.md > /dev/null 2> $O/gemm_shapes.err
python3 $R/profiles/scmn_bench.py > $O/scmn.json 2> $O/scmn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_scmn -o run -- python3 $R/profiles/scmn_bench.py > /dev/null 2>> $O/prof.err
python3 $R/profiles/tn_ws_bench.py --md $O/tn_ws.md > /dev/null 2> $O/tn_ws.err
ls -la $O
