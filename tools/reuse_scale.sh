# evaluation-reuse leg against the number of concurrent games, then config 2 played to completion with the engine's defaults
set -e
for G in 4096 8192 16384; do
  python bench.py --games $G --reuse-evaluations 1 --steps 6 --warmup 5 --reuse-steps 0 --secondary-nn none --no-cpu-baseline > gpurun_out/reuse_games_$G.json 2> gpurun_out/reuse_games_$G.err
  python -c "import json;d=json.load(open('gpurun_out/reuse_games_$G.json'));print('games', $G, d['value'], d['ms_per_step'], d['expansions_per_s'])"
done
python tools/config2_full.py --reuse 1 --book 8 --out gpurun_out/config2_reuse_final.json > gpurun_out/config2_reuse_final.log 2>&1
tail -3 gpurun_out/config2_reuse_final.log
