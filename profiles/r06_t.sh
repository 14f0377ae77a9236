#!/bin/bash
# A/B on one box: k_entropy_bits table builds without the per-part exchange barrier (D, this build) against the build of profiles/r06_s "B"
O=gpurun_out/r06t; mkdir -p $O
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
for v in xb D; do
  lib=$P/libtsfresh_amd_$v.so; [ $v = D ] && lib=$P/libtsfresh_amd.so
  run ${v}_1024_$rep $lib --n-series 100000 --length 1024
  run ${v}_256_$rep $lib --n-series 125000 --length 256
  run ${v}_2048_$rep $lib --n-series 20000 --ragged 1025:2048
done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_entropy_hbits.py -x -q -m gpu -k "entropy or golden or propert" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python profiles/fuzz_parity.py 30 2024 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
