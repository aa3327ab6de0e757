# A/B of LockstepSearch's simulations-per-graph on one box: evaluation-reuse leg as the main leg, then the headline
set -e
for U in 1 8 1 8 16 4; do
  YY_GRAPH_UNROLL=$U python bench.py --reuse-evaluations 1 --steps 6 --warmup 4 --reuse-steps 0 --secondary-nn none --no-cpu-baseline > gpurun_out/reuse_unroll_$U.json 2> gpurun_out/reuse_unroll_$U.err
  python -c "import json;d=json.load(open('gpurun_out/reuse_unroll_$U.json'));print('reuse unroll', $U, d['value'], d['ms_per_step'])"
done
for U in 1 8; do
  YY_GRAPH_UNROLL=$U python bench.py --steps 4 --warmup 2 --reuse-steps 0 --secondary-nn none --no-cpu-baseline > gpurun_out/head_unroll_$U.json 2> gpurun_out/head_unroll_$U.err
  python -c "import json;d=json.load(open('gpurun_out/head_unroll_$U.json'));print('headline unroll', $U, d['value'], d['ms_per_step'])"
done
