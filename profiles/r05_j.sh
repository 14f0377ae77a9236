#!/bin/bash
# round 5, call J: workgroup size of k_spectral with the rebuilt chirp-z transform, and the crossover
export TMPDIR=/tmp
O=gpurun_out/r05_j; rm -rf $O; mkdir -p $O
run() { # label, ragged, n, env
  env $4 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --params efficient --ragged $2 --n-series $3 2>/dev/null | tail -1 > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('$1', '$2', '$4', round(d['ms_per_step'],2), 'k_spectral', round(d['kernel_ms']['k_spectral'],2), d.get('parity_sample'))" | tee -a $O/quick.txt
}
for nt in 256 512 1024; do run cfg5 4096:8192 5000 "TSFA_NT_2=$nt"; done
for nt in 128 256 512; do run mid 2049:4096 5000 "TSFA_BLUESTEIN_MIN=1537 TSFA_NT_2=$nt"; done
for nt in 64 128 256; do run low 1025:2048 10000 "TSFA_BLUESTEIN_MIN=1281 TSFA_NT_2=$nt"; done
for m in 1281 1793 2049; do run low 1025:2048 10000 "TSFA_BLUESTEIN_MIN=$m TSFA_NT_2=256"; done
