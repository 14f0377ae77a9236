#!/bin/bash
# round 5, call K: chirp-z transform with the crossover at 1281 and its workgroup sizes: the gpu tests that touch spectral
# columns, then the shapes
export TMPDIR=/tmp
O=gpurun_out/r05_k; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_param_sweep.py tests/test_roll.py -m gpu -q -x -k "chirp or golden or config or longer_than_lds or ragged or sweep or roll or every_cell" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
run() { # label, params, ragged-or-length args, n
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --params $2 $3 --n-series $4 2>/dev/null | tail -1 > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('$1', '$2', '$3', $4, round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['kernel_ms'].items()}, d.get('parity_sample'))" | tee -a $O/quick.txt
}
run cfg5 efficient "--ragged 4096:8192" 5000
run cfg5full efficient "--ragged 4096:8192" 12500
run mid efficient "--ragged 2049:4096" 5000
run low efficient "--ragged 1025:2048" 10000
run cfg2 efficient "--length 1024" 10000
run odd1000 comprehensive "--length 1000" 20000
