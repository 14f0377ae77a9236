#!/bin/bash
# round 5, call M: chirp-z transform, crossover 1281 (odd) / 897 (even): the gpu tests that touch spectral columns, the shapes
export TMPDIR=/tmp
O=gpurun_out/r05_m; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_param_sweep.py tests/test_roll.py tests/test_frames.py -m gpu -q -x -k "chirp or golden or config or longer_than_lds or ragged or sweep or roll or frame" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
run() { # label, params, args, n
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --params $2 $3 --n-series $4 2>/dev/null | tail -1 > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('$1', '$2', '$3', $4, round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['kernel_ms'].items()}, d.get('parity_sample'))" | tee -a $O/quick.txt
}
run cfg5 efficient "--ragged 4096:8192" 5000
run mid efficient "--ragged 2049:4096" 5000
run low efficient "--ragged 1025:2048" 10000
run r513 efficient "--ragged 513:1024" 20000
run len1000 efficient "--length 1000" 20000
run len1000c comprehensive "--length 1000" 20000
run headline comprehensive "--length 1024" 100000
