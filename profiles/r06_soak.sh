#!/bin/bash
# device soak of the final build: four fuzz modes, other seeds than the measurement set's
O=gpurun_out/r06soak; mkdir -p $O
sha256sum tsfresh_amd/libtsfresh_amd.so | cut -c1-16 > $O/lib_sha16.txt
timeout 900 python profiles/fuzz_parity.py 60 90210 > $O/std.log 2>&1; tail -1 $O/std.log
TSFA_FUZZ_PARAMS=random timeout 900 python profiles/fuzz_parity.py 40 8086 > $O/random_params.log 2>&1; tail -1 $O/random_params.log
TSFA_FUZZ_MAXLENS=200,1100,1600,2048,2500,3900,4096,5000 timeout 1500 python profiles/fuzz_parity.py 24 6502 > $O/long.log 2>&1; tail -1 $O/long.log
TSFA_FUZZ_EXTREME=1 timeout 600 python profiles/fuzz_parity.py 30 68000 > $O/extreme.log 2>&1; tail -1 $O/extreme.log
grep -h "mismatches [1-9]\|UNWRITTEN" $O/*.log | cut -c1-300 | head
