# network.G_SPLIT_WG on the evaluation-reuse leg (main leg here), one box
set -e
for S in 512 256 384 192 512 320 640; do
  python bench.py --reuse-evaluations 1 --split-wg $S --steps 6 --warmup 4 --reuse-steps 0 --secondary-nn none --no-cpu-baseline > gpurun_out/reuse_split_$S.json 2> gpurun_out/reuse_split_$S.err
  python -c "import json;d=json.load(open('gpurun_out/reuse_split_$S.json'));print('split', $S, d['value'], d['ms_per_step'])"
done
