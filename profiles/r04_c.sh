#!/bin/bash
# round 4, call C: the long fixtures on the device, config 5 at 2 000 series, the fill audit
export TMPDIR=/tmp
O=gpurun_out/r04_c; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "long or config5" > $O/pytest_long.log 2>&1; tail -5 $O/pytest_long.log
timeout 900 python profiles/fill_audit.py > $O/fill_audit.json 2> $O/fill_audit.err; tail -2 $O/fill_audit.err
