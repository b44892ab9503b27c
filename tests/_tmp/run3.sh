cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5 || exit 1
timeout -k 10 120 python3 bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-140
timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-1000
timeout -k 10 200 python3 tests/split_probe.py 2>&1 | grep -E "rank 0 of"
