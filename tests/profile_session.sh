#!/bin/bash
# Profiling session on the GPU box (a helper, not a test): the bench lines and the rocprofv3 passes that
# tests/make_profiles.py turns into the summaries under profiles/.  Every pass is a run of its own with a
# limit of its own; counters are collected without any trace option.
#   bash tests/profile_session.sh gpurun_out/session && python3 tests/make_profiles.py gpurun_out/session r2
set -e
out=${1:-gpurun_out/session}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
timeout -k 10 200 python3 bench.py > "$out/r1_bench.json" 2> "$out/bench.err"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 > "$out/r1_bench_steps20_warmup5.json" 2>> "$out/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 bench.py > "$out/r1_bench_under_rocprof.json" 2> "$out/kt.err"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$out/sq" -o sq -- python3 bench.py --no-cpu-baseline --jobs 0 > "$out/sq.json" 2> "$out/sq.err"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fs" -o fs -- python3 bench.py --no-cpu-baseline --jobs 0 > "$out/fs.json" 2> "$out/fs.err"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/ws" -o ws -- python3 bench.py --no-cpu-baseline --jobs 0 > "$out/ws.json" 2> "$out/ws.err"
# flatten: make_profiles.py looks for the csv files directly under kt/ sq/ fs/ ws/
for d in kt sq fs ws; do find "$out/$d" -mindepth 2 -name '*.csv' -exec mv {} "$out/$d/" \; ; done
ls "$out" "$out/kt" | head -40
