cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_a -o a -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/prof_a.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_a/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(10), r['Percentage'].rjust(7))
PY
