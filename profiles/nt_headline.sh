#!/bin/bash
# workgroup size per family on the headline workload (TSFA_NT_<family index> hook of tsfa_api.cpp), one change at a time
export TMPDIR=/tmp
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))"; }
echo "base"; run
for kv in "$@"; do echo "== $kv"; env $kv bash -c "$(declare -f run); run"; done
