#!/bin/bash
# round 6 device fuzz: the row form (series <= 256 samples are a third of every batch), the bit-matrix sweep beyond 4096 samples,
# the chunked histograms, random parameters
O=gpurun_out/r06fuzz; mkdir -p $O
timeout 900 python profiles/fuzz_parity.py 60 601 > $O/fuzz_std.log 2>&1; tail -1 $O/fuzz_std.log
TSFA_FUZZ_PARAMS=random timeout 900 python profiles/fuzz_parity.py 60 602 > $O/fuzz_random_params.log 2>&1; tail -1 $O/fuzz_random_params.log
TSFA_FUZZ_MAXLENS=200,256,5000,7000 timeout 1500 python profiles/fuzz_parity.py 24 603 > $O/fuzz_long.log 2>&1; tail -1 $O/fuzz_long.log
TSFA_FUZZ_EXTREME=1 TSFA_FUZZ_MAXLENS=40,300,1024 timeout 1200 python profiles/fuzz_parity.py 100 61 > $O/fuzz_extreme.log 2>&1; tail -1 $O/fuzz_extreme.log
grep -h "mismatches [1-9]\|UNWRITTEN" $O/*.log | cut -c1-400 | head -20
