#!/bin/bash
# A/B on one box: k_entropy_bits with the upper range ends by inversion (A), the closing logarithms re-dealt (B), neither (C), both (D)
O=gpurun_out/r06s; mkdir -p $O
run() { # name lib args...
  name=$1; lib=$2; shift; shift
  TSFA_LIB=$lib python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e "$@" > $O/$name.json 2>$O/$name.err
  python - <<PY
import json
d=json.loads(open('$O/$name.json').read().strip().split('\n')[-1])
print('$name', round(d['ms_per_step'],3), 'k_entropy', round(d['kernel_ms']['k_entropy'],3), d.get('parity_sample'))
PY
}
P=$PWD/tsfresh_amd
for rep in 1 2; do
for v in xc xb xa D; do
  lib=$P/libtsfresh_amd_$v.so; [ $v = D ] && lib=$P/libtsfresh_amd.so
  run ${v}_1024_$rep $lib --n-series 100000 --length 1024
  run ${v}_256_$rep $lib --n-series 125000 --length 256
done
done
