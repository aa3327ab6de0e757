#!/bin/bash
# Round-2 profiles on the MI355X box (run from the repo root through gpurun): kernel-trace stats of the bench command and the
# PMC passes of the headline evaluator kernel (separate passes: --pmc never together with a trace domain).
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"
O="$R/gpurun_out/prof_r02"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
K="k_tower_h3"
PY="$R/tools/tower_micro_h3.py"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --secondary-nn none --oversubscribe 0 --reuse-steps 0 > "$O/bench_stats_run.json" 2> "$O/bench_stats_run.err" || exit 1
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo "$pass" | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $pass --kernel-include-regex "$K" --output-format csv -d "$O/pmc_$tag" -- python3 "$PY" 4096 6 r > "$O/pmc_$tag.log" 2>&1 || exit 1
done
# the tree kernel: HBM bytes per launch of k_mcts in the benchmark workload itself (eager launches: a graph replay is one dispatch)
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
    timeout -k 10 300 rocprofv3 --pmc $pass --kernel-include-regex "k_mcts" --output-format csv -d "$O/pmc_mcts_$pass" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-graph --no-cpu-baseline --secondary-nn none --reuse-steps 0 > "$O/pmc_mcts_$pass.log" 2>&1 || exit 1
done
echo profiles done
# the other board sizes of the float32-accurate tower (BASELINE configs 1 and 4): kernel-trace summary of the dense 4096-board launch
for RR in 6 12; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tower_$RR" -- python3 "$PY" 4096 10 r $RR > "$O/tower_$RR.log" 2>&1 || exit 1
done
echo board sizes done
