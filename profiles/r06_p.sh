#!/bin/bash
# number_cwt_peaks beyond ~1400 samples: the SNR filter by window counts (cwt_filter_counts) instead of argsort + order walk
O=gpurun_out/r06p; mkdir -p $O
timeout 1200 python -m pytest tests/test_cwt_peaks_long.py tests/test_cwt_peaks_mfma.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { # name, args...
  name=$1; shift
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $O/$name.json 2>$O/$name.err
  python - <<PY
import json
d=json.loads(open('$O/$name.json').read().strip().split('\n')[-1])
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()})
PY
}
run cfg4 --n-series 5000 --ragged 4096:8192 --params efficient
run c4096 --n-series 10000 --length 4096
run h1024 --n-series 100000 --length 1024
run c2048 --n-series 20000 --ragged 1025:2048
TSFA_FUZZ_MAXLENS=200,256,5000,7000 timeout 1500 python profiles/fuzz_parity.py 24 613 > $O/fuzz_long.log 2>&1; tail -1 $O/fuzz_long.log
grep -h "mismatches [1-9]\|UNWRITTEN" $O/*.log | cut -c1-400 | head -20
