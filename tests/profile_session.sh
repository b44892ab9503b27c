#!/bin/bash
# Profiling session on the GPU box (a helper, not a test): the bench lines and the rocprofv3 passes that
# tests/make_profiles.py turns into the summaries under profiles/.  Every pass is a run of its own with a
# limit of its own; counters are collected without any trace option.
#   bash tests/profile_session.sh gpurun_out/session && python3 tests/make_profiles.py gpurun_out/session r3
#   bash tests/profile_session.sh gpurun_out/session configs      (the other configurations only, a call of its own)
set -e
out=${1:-gpurun_out/session}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
if [ "$2" != "configs" ]; then
timeout -k 10 200 python3 bench.py > "$out/line_bench.json" 2> "$out/bench.err"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 > "$out/line_bench_steps20_warmup5.json" 2>> "$out/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 bench.py > "$out/line_bench_under_rocprof.json" 2> "$out/kt.err"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$out/sq" -o sq -- python3 bench.py --no-cpu-baseline --jobs 0 > "$out/sq.json" 2> "$out/sq.err"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fs" -o fs -- python3 bench.py --no-cpu-baseline --jobs 0 > "$out/fs.json" 2> "$out/fs.err"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/ws" -o ws -- python3 bench.py --no-cpu-baseline --jobs 0 > "$out/ws.json" 2> "$out/ws.err"
# flatten: make_profiles.py looks for the csv files directly under kt/ sq/ fs/ ws/
for d in kt sq fs ws; do find "$out/$d" -mindepth 2 -name '*.csv' -exec mv {} "$out/$d/" \; ; done
ls "$out" "$out/kt" | head -40
exit 0
fi
# the other configurations and the per-rank shares of an 8-GPU job, one bench line each
timeout -k 10 200 python3 bench.py --size 512 --steps 20 --warmup 5 --jobs 10 --no-cpu-baseline > "$out/cfg_dragon_512.json" 2>> "$out/bench.err"
timeout -k 10 200 python3 bench.py --scene cornell_box.scene --size 512 --steps 20 --warmup 5 --jobs 10 --no-cpu-baseline > "$out/cfg_box_512.json" 2>> "$out/bench.err"
timeout -k 10 200 python3 bench.py --scene cornell_sphere.scene --size 512 --steps 20 --warmup 5 --jobs 10 --no-cpu-baseline > "$out/cfg_sphere_512.json" 2>> "$out/bench.err"
timeout -k 10 200 python3 bench.py --scene cornell_sphere.scene --size 1024 --steps 20 --warmup 5 --jobs 10 --no-cpu-baseline > "$out/cfg_sphere_1024.json" 2>> "$out/bench.err"
timeout -k 10 200 python3 bench.py --tree median --steps 20 --warmup 5 --jobs 10 --no-cpu-baseline > "$out/cfg_tree_median.json" 2>> "$out/bench.err"
timeout -k 10 300 python3 bench.py --photons 2000000 500000 --steps 20 --warmup 5 --no-cpu-baseline --jobs 10 > "$out/cfg_2m_500k.json" 2>> "$out/bench.err"
timeout -k 10 400 python3 bench.py --synthetic 2236 --size 4096 --photons 16000000 4000000 --no-cpu-baseline --jobs 0 --steps 8 --warmup 5 > "$out/cfg_synthetic_field.json" 2>> "$out/bench.err"
# two ranks sharing the one GPU (gloo moves the exchange through host memory): the launcher and ranks_seen
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --no-cpu-baseline --jobs 0 > "$out/cfg_two_ranks_gloo.json" 2>> "$out/bench.err"
timeout -k 10 300 python3 bench.py --gpus 2 --driver group --share-gpus --no-cpu-baseline --jobs 0 > "$out/cfg_two_ranks_group.json" 2>> "$out/bench.err"
ls "$out" | head -40
