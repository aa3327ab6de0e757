# lanes at small batches (config 5 plays 512 games per iteration), evaluation-reuse leg as the main leg, one box
set -e
for G in 512 1024; do
for L in 1 2 4; do
  python bench.py --games $G --lanes $L --reuse-evaluations 1 --steps 6 --warmup 4 --reuse-steps 0 --secondary-nn none --no-cpu-baseline > gpurun_out/small_g${G}_l$L.json 2> gpurun_out/small_g${G}_l$L.err
  python -c "import json;d=json.load(open('gpurun_out/small_g${G}_l$L.json'));print('games', $G, 'lanes', $L, round(d['value'],1), round(d['ms_per_step'],1))"
done
done
