#!/bin/bash
# cost experiment: k_entropy_bits without its sample sort (identity order: wrong results, timing only)
export TMPDIR=/tmp
cd tsfresh_amd/csrc
F="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -disable-machine-licm"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -DTSFA_EXPERIMENT_NO_ENTB_SORT -c tsfa_kernels.hip -o /tmp/k_nosort.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/k_nosort.o tsfa_kernels_long.o tsfa_api.o tsfa_relevance.o tsfa_pack.o -lpthread -o /tmp/libtsfresh_amd_nosort.so
cd ../..
for lib in "" /tmp/libtsfresh_amd_nosort.so; do
  TSFA_LIB=$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()})"
done
