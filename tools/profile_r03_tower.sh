#!/bin/bash
# Round-3 PMC passes of the evaluator tower kernels on the MI355X box (through gpurun, from the repo root): the general
# 16x16x32 kernel (k_tower_g) and the 32x32x16 register-ring kernel (k_tower_h3r) on the same box, separate passes per group.
# usage: tools/profile_r03_tower.sh <tag> <kernel regex> <micro args...>     e.g.  g8 k_tower_g 4096 6 g 8
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"
TAG="$1"; K="$2"; shift 2
O="$R/gpurun_out/prof_r03/$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
PY="$R/tools/tower_micro_h3.py"
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo "$pass" | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $pass --kernel-include-regex "$K" --output-format csv -d "$O/pmc_$tag" -- python3 "$PY" "$@" > "$O/pmc_$tag.log" 2>&1 || exit 1
done
echo "profiles $TAG done"
