# network.G_SPLIT_WG at the other BASELINE shapes (evaluation-reuse leg as the main leg), one box
set -e
for S in 512 256 512 256; do
  python bench.py --rows 6 --cols 6 --sims 25 --reuse-evaluations 1 --split-wg $S --steps 20 --warmup 6 --reuse-steps 0 --secondary-nn none --no-cpu-baseline > gpurun_out/split6_$S.json 2> gpurun_out/split6_$S.err
  python -c "import json;d=json.load(open('gpurun_out/split6_$S.json'));print('6x6/25 split', $S, d['value'], d['ms_per_step'])"
done
for S in 512 256; do
  python bench.py --rows 12 --cols 12 --sims 1600 --reuse-evaluations 1 --split-wg $S --steps 2 --warmup 2 --reuse-steps 0 --secondary-nn none --no-cpu-baseline > gpurun_out/split12_$S.json 2> gpurun_out/split12_$S.err
  python -c "import json;d=json.load(open('gpurun_out/split12_$S.json'));print('12x12/1600 split', $S, d['value'], d['ms_per_step'])"
done
