#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_d
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lab profiles/lab/minimal_fused_lab.hip && /tmp/lab > $O/lab.txt 2>&1
cat $O/lab.txt
run() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '|', round(d['ms_per_step'],2), 'ms |', d.get('parity_sample'), {k: round(v,2) for k,v in d['kernel_ms'].items()})"; }
{
run
TSFA_NO_PERM_SHARE=1 run
} > $O/perm_ab.txt 2>&1
cat $O/perm_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
