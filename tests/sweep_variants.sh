#!/bin/bash
# Runs the bench line once per variant library under sweep_libs/ (built with other values of the tuning flags,
# e.g. -DTK_WALK_REST_SHIFT / -DTK_LEAF_SHARE) and once with the library as built: a helper for sweeps, not a test.
#   bash tests/sweep_variants.sh gpurun_out/sweep [bench arguments]
set -e
out=${1:-gpurun_out/sweep}; shift || true
mkdir -p "$out"
lib=pathtracer_amd/lib/libtkhip.so
cp $lib "$out/libtkhip_base.so.keep"
run() { timeout -k 10 200 python3 bench.py --no-cpu-baseline --jobs 0 --steps 20 --warmup 5 "$@"; }
run "$@" > "$out/base.json" 2> "$out/base.err"
for v in sweep_libs/libtkhip_*.so; do
  n=$(basename $v .so); n=${n#libtkhip_}
  cp $v $lib
  run "$@" > "$out/$n.json" 2> "$out/$n.err" || echo "$n failed"
done
cp "$out/libtkhip_base.so.keep" $lib
rm -f "$out/libtkhip_base.so.keep"
run "$@" > "$out/base2.json" 2> "$out/base2.err"
for f in "$out"/*.json; do echo "$(basename $f .json) $(python3 -c "import json,sys; print(round(json.loads(open('$f').read().strip().splitlines()[-1])['ms_per_step'],3))")"; done
