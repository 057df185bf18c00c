#!/bin/bash
# Developer helper (GPU box): run bench.py through its option paths (--separate, --fast, --stencil 9, other sizes) with short timed regions -- a does-it-run check, not a measurement.
cd $GRAFT_REPO_ROOT
for a in "--separate" "--fast" "--f64" "--stencil 9" "--n 512 --batch 64" "--n 256 --batch 16 --distinct 2"; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline $a 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$a', round(d['value']/1e10,3), d['config']['workload'][:70], d['roofline']['kernel'], round(d['roofline']['frac'],3))"
done
