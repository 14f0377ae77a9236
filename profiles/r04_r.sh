#!/bin/bash
# round 4, call R: phase clocks of permutation_entropy, all dimensions from one sweep vs one by one
export TMPDIR=/tmp
O=gpurun_out/r04_r; rm -rf $O; mkdir -p $O
export TSFA_LIB=$PWD/tsfresh_amd/libtsfresh_amd_ticks.so
timeout 300 python profiles/phase_ticks.py 2> $O/fused.err | grep -i "permutation" > $O/fused.md
TSFA_NO_PE_FUSED=1 timeout 300 python profiles/phase_ticks.py 2> $O/onebyone.err | grep -i "permutation" > $O/onebyone.md
echo fused; cat $O/fused.md; echo one by one; cat $O/onebyone.md
