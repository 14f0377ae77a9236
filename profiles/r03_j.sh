#!/bin/bash
# rocprofv3 kernel stats of the two new kernels' workloads: Comprehensive 10 000 x 4096 (k_entropy_bits<T, 3>) and
# MinimalFCParameters 100 000 x 1024 (k_stream)
export TMPDIR=/tmp
O=gpurun_out/r03_j; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p1 -o p -- python bench.py --no-cpu-baseline --no-e2e --n-series 10000 --length 4096 > $O/b1.json 2> $O/e1.log
DB=$(ls $O/p1/*/*.db $O/p1/*.db 2>/dev/null | head -1); [ -n "$DB" ] && python profiles/summarize_rocpd.py $DB "r03_j: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-e2e --n-series 10000 --length 4096" > $O/kernel_stats_4096.md
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p2 -o p -- python bench.py --no-cpu-baseline --no-e2e --params minimal --steps 20 --warmup 3 > $O/b2.json 2> $O/e2.log
DB=$(ls $O/p2/*/*.db $O/p2/*.db 2>/dev/null | head -1); [ -n "$DB" ] && python profiles/summarize_rocpd.py $DB "r03_j: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-e2e --params minimal --steps 20 --warmup 3" > $O/kernel_stats_minimal.md
head -14 $O/kernel_stats_4096.md; head -8 $O/kernel_stats_minimal.md
rm -rf $O/p1 $O/p2
