set -e
for M in 2 4 8 2; do
python tools/config2_full.py --single --reuse 1 --book 8 --hint-mult $M --out gpurun_out/c2_hint_$M.json 2>&1 | grep wall_s | cut -c100-200
done
