cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/prof_b
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b -o k --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_b.log 2>&1)
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_b/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:40]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), ('%.2f%%'%(100*float(r['TotalDurationNs'])/tot)).rjust(7))
print('total kernel ms per forward', tot/1e6/8, 'n kernels', sum(int(r['Calls']) for r in rows)/8)
PY
cp $(ls gpurun_out/prof_b/*kernel_stats.csv | head -1) gpurun_out/r02_mid_kernel_stats.csv
find gpurun_out/prof_b -name '*kernel_trace*' -size +20M -delete
