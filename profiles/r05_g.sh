#!/bin/bash
# round 5, call G: the rebuilt chirp-z transform (fam_spectral.h) on non-power-of-two lengths -- long goldens, then the
# kernel time on ragged shapes for several crossovers against the Goertzel sweep
export TMPDIR=/tmp
O=gpurun_out/r05_i; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden and long or config5 or longer_than_lds or ragged" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
run() { # label, ragged, n, env
  env $4 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --params efficient --ragged $2 --n-series $3 2>/dev/null | tail -1 > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('$1', '$2', '$4', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['kernel_ms'].items()}, d.get('parity_sample'))" | tee -a $O/quick.txt
}
run cfg5 4096:8192 5000 TSFA_NO_BLUESTEIN=1
run cfg5 4096:8192 5000 TSFA_BLUESTEIN_MIN=4097
run mid 2049:4096 5000 TSFA_NO_BLUESTEIN=1
run mid 2049:4096 5000 TSFA_BLUESTEIN_MIN=2049
run low 1025:2048 10000 TSFA_NO_BLUESTEIN=1
run low 1025:2048 10000 TSFA_BLUESTEIN_MIN=1025
