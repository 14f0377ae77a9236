#!/bin/bash
# round 3, first GPU pass: parity tests, the headline bench, the walk / offset variants, other shapes
export TMPDIR=/tmp
O=gpurun_out/r03_a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
for v in "--walk" "--offset 1000" "--offset 100000" "--walk --offset 1e6"; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e $v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), d.get('parity_sample'), {k: round(v,2) for k,v in d['kernel_ms'].items()})"
done > $O/variants.txt 2>&1
cat $O/variants.txt
