#!/bin/bash
# round 5, call C (third form: K loops unrolled per width, no per-step clamp): number_cwt_peaks' Ricker convolutions on the float64 matrix cores (cwt_rows_mfma) -- parity against the
# VALU tiles and the oracle, every fixture pair through the C-ABI, and the A/B of the kernel time on the headline shape
export TMPDIR=/tmp
O=gpurun_out/r05_d; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_cwt_peaks_mfma.py -m gpu -q -x > $O/pytest_mfma.log 2>&1; echo "rc=$?" >> $O/pytest_mfma.log; tail -15 $O/pytest_mfma.log
echo skipped-golden
for v in 0 1; do
  TSFA_NO_CWT_MFMA=$v timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/bench_nomfma$v.json
  python -c "import json; d=json.load(open('$O/bench_nomfma$v.json')); print('TSFA_NO_CWT_MFMA=$v', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['kernel_ms'].items()}, d.get('parity_sample'))" | tee -a $O/quick.txt
done
