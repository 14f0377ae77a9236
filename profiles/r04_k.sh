#!/bin/bash
# round 4, call K: a low-priority side lane for the latency-bound families (TSFA_PAIR)
export TMPDIR=/tmp
O=gpurun_out/r04_k; rm -rf $O; mkdir -p $O
q() { timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-e2e $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$1', round(d['ms_per_step'],2), d.get('parity_sample'))"; }
{
q "one lane"
TSFA_PAIR=seq q "side: seq"
TSFA_PAIR=seq,spectral q "side: seq,spectral"
TSFA_PAIR=seq,spectral,cwt q "side: seq,spectral,cwt"
TSFA_PAIR=seq,spectral,cwt,trend q "side: seq,spectral,cwt,trend"
TSFA_PAIR=seq,spectral,cwt,trend,ar q "side: ... + ar"
TSFA_PAIR=seq,cwt q "side: seq,cwt"
TSFA_PAIR=seq q "side: seq  (walk)" --walk
q "one lane   (walk)" --walk
TSFA_PAIR=seq,spectral,cwt,trend q "side: 4 fam (256)" "--n-series 125000 --length 256"
q "one lane (256)" "--n-series 125000 --length 256"
} > $O/pair.txt 2>&1
cat $O/pair.txt
