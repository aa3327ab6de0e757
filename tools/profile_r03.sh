#!/bin/bash
# Round-3 profiles on the MI355X box (run from the repo root through gpurun): kernel-trace stats of the bench command (headline
# leg, then the evaluation-reuse leg as the main leg), the PMC passes of the tree kernel inside the benchmark workload, and the
# bench lines of BASELINE configs 1 and 4.  (The tower's PMC passes: tools/profile_r03_tower.sh.)  --pmc never with a trace domain.
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"
O="$R/gpurun_out/prof_r03"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --secondary-nn none --reuse-steps 0 > "$O/bench_stats_run.json" 2> "$O/bench_stats_run.err" || exit 1
echo "bench stats done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_reuse" -- python3 "$R/bench.py" --steps 6 --warmup 4 --no-cpu-baseline --secondary-nn none --reuse-steps 0 --reuse-evaluations 1 > "$O/bench_reuse_stats_run.json" 2> "$O/bench_reuse_stats_run.err" || exit 1
echo "reuse stats done"
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
    timeout -k 10 300 rocprofv3 --pmc $pass --kernel-include-regex "k_mcts" --output-format csv -d "$O/pmc_mcts_$pass" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-graph --no-cpu-baseline --secondary-nn none --reuse-steps 0 > "$O/pmc_mcts_$pass.log" 2>&1 || exit 1
done
echo "mcts pmc done"
cd "$R"
timeout -k 10 300 python3 bench.py --rows 6 --cols 6 --sims 25 --steps 20 --warmup 4 --no-cpu-baseline --secondary-nn none > "$O/bench_6x6_25.json" 2> "$O/bench_6x6_25.err" || exit 1
timeout -k 10 500 python3 bench.py --rows 12 --cols 12 --sims 1600 --steps 2 --warmup 1 --no-cpu-baseline --secondary-nn none --reuse-steps 3 > "$O/bench_12x12_1600.json" 2> "$O/bench_12x12_1600.err" || exit 1
echo "configs 1 and 4 done"
