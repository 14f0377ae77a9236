#!/bin/bash
# round 5, call L: crossover below 1281 (typical series lengths are not powers of two: 1000, 720, 500 ...)
export TMPDIR=/tmp
O=gpurun_out/r05_l; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chirp" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
run() { # label, args, n, env
  env $4 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --params efficient $2 --n-series $3 2>/dev/null | tail -1 > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('$1', '$2', $3, '$4', round(d['ms_per_step'],2), 'k_spectral', round(d['kernel_ms']['k_spectral'],2), d.get('parity_sample'))" | tee -a $O/quick.txt
}
for m in 1281 769 513 257; do run len1000 "--length 1000" 20000 "TSFA_BLUESTEIN_MIN=$m"; done
for m in 1281 513; do run len1000nt128 "--length 1000" 20000 "TSFA_BLUESTEIN_MIN=$m TSFA_NT_2=128"; done
for m in 1281 513 257; do run len500 "--length 500" 40000 "TSFA_BLUESTEIN_MIN=$m"; done
for m in 1281 257; do run len300 "--length 300" 40000 "TSFA_BLUESTEIN_MIN=$m"; done
for m in 1281 769 513; do run r513 "--ragged 513:1024" 20000 "TSFA_BLUESTEIN_MIN=$m"; done
