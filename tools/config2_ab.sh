set -e
python tools/config2_full.py --reuse 1 --book 8 --out gpurun_out/c2_final_reuse.json 2>&1 | grep wall_s | cut -c1-220
python tools/config2_full.py --rows 12 --games 1024 --slots 1024 --sims 1600 --single --out gpurun_out/c4_final_reuse.json 2>&1 | grep wall_s | cut -c1-220
