#!/bin/bash
# round 5: the parity fuzz on the device (random calculator subsets / random parameters x random ragged batches against the
# oracle), every plan with the sentinel pre-fill.   usage: bash profiles/r05_fuzz.sh TAG
TAG=${1:-r05_y}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
( timeout 500 python profiles/fuzz_parity.py 36 ${SEED:-5101}; TSFA_FUZZ_PARAMS=random timeout 500 python profiles/fuzz_parity.py 36 $((${SEED:-5101}+1));
  TSFA_FUZZ_MAXLENS="300,1024,2500,4500,9000" timeout 500 python profiles/fuzz_parity.py 12 $((${SEED:-5101}+2)) ) > $O/fuzz_gpu.log 2>&1
grep -c "^round" $O/fuzz_gpu.log; grep "TOTAL\|UNWRITTEN" $O/fuzz_gpu.log
