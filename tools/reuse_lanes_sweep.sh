set -e
for L in 2 3 4 6; do
  python bench.py --reuse-evaluations 1 --lanes $L --steps 6 --warmup 3 --reuse-steps 0 --secondary-nn none --no-cpu-baseline > gpurun_out/reuse_lanes_$L.json 2> gpurun_out/reuse_lanes_$L.err
  python -c "import json;d=json.load(open('gpurun_out/reuse_lanes_$L.json'));print($L, d['value'], d['ms_per_step'])"
done
