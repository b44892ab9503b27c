#!/bin/bash
# Instruction cache counters of the kernels of the bench iteration (a helper, not a test): one rocprofv3 --pmc pass,
# a run of its own without any trace option; the csv lands under gpurun_out/r3_l/a.
set -e
out=gpurun_out/r3_l
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_IFETCH_LEVEL --output-format csv -d $out/a -o a -- python3 bench.py --no-cpu-baseline --jobs 0 --steps 5 --warmup 3 > $out/a.json 2> $out/a.err
find "$out/a" -mindepth 2 -name '*counter_collection.csv' -exec mv {} "$out/a/" \;
ls $out/a
