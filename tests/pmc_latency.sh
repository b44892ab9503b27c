#!/bin/bash
# Latency and cache counters of the kernels of the bench iteration (a helper, not a test): three rocprofv3 --pmc passes,
# each a run of its own without any trace option; the csv files land under gpurun_out/r3_k/{a,b,c}.
set -e
out=gpurun_out/r3_k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $out/a -o a -- python3 bench.py --no-cpu-baseline --jobs 0 --steps 5 --warmup 3 > $out/a.json 2> $out/a.err
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $out/b -o b -- python3 bench.py --no-cpu-baseline --jobs 0 --steps 5 --warmup 3 > $out/b.json 2> $out/b.err
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $out/c -o c -- python3 bench.py --no-cpu-baseline --jobs 0 --steps 5 --warmup 3 > $out/c.json 2> $out/c.err
for d in a b c; do find "$out/$d" -mindepth 2 -name '*counter_collection.csv' -exec mv {} "$out/$d/" \; ; done
ls $out/a $out/b $out/c
