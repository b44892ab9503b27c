cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in sh3 sh4; do
cp tests/_tmp/$v/libtkhip.so pathtracer_amd/lib/libtkhip.so
echo "== $v"
timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-140 || exit 1
timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-140 || exit 1
done
