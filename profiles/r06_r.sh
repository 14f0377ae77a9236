#!/bin/bash
# k_entropy_bits: upper range ends as the inverse of the lower ones (no second set of bisections), closing logarithms off the
# critical wavefront; full gpu suite of this build
O=gpurun_out/r06r; mkdir -p $O
run() { # name, args...
  name=$1; shift
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $O/$name.json 2>$O/$name.err
  python - <<PY
import json
d=json.loads(open('$O/$name.json').read().strip().split('\n')[-1])
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))
PY
}
run h1024 --n-series 100000 --length 1024
run h256 --n-series 125000 --length 256
run h1000 --n-series 20000 --length 1000
run walk --walk
timeout 900 python profiles/fuzz_parity.py 40 1311 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
grep -h "mismatches [1-9]\|UNWRITTEN" $O/*.log | cut -c1-400 | head -20
