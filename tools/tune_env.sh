run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['value']/1e10,3), {k:round(v['avg_launch_ms'],4) for k,v in d['roofline']['all_kernels'].items()})"; }
for round in 1 2; do
  run base
  for g in 256 384 768 1024; do NNS_SPEC_GRID=$g run grid$g; done
  for r in 8 16 64; do NNS_MARCH_R=$r run R$r; done
done
