#!/bin/bash
# round 4, call U: k_perm's sums from integer counts (independent of the thread count): the multi-device bit-identity test, the
# permutation_entropy tests, one headline step
export TMPDIR=/tmp
O=gpurun_out/r04_u; rm -rf $O; mkdir -p $O
timeout 80 python -m pytest tests -m gpu -q -k "perm or several_devices" > $O/pytest_some.log 2>&1; echo "pytest rc=$?" >> $O/pytest_some.log; tail -3 $O/pytest_some.log
timeout 60 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/bench_quick.json; python -c "import json; d=json.load(open('$O/bench_quick.json')); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))"
