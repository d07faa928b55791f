export TMPDIR=/tmp
i=0
for wv in "2,3,4,5" "3,3,4,5"; do
  i=$((i+1))
  ./build.sh "-DTWX_UKW_WV=$wv" >/dev/null 2>&1
  rocprofv3 --kernel-trace --stats -d gpurun_out/ab_w$i -o s --output-format csv -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-daily 2>/dev/null | tail -1 > gpurun_out/ab_w$i.json
  echo "== UKW_WV=$wv"; python3 -c "
import json; d=json.load(open('gpurun_out/ab_w$i.json')); print(d['timing_ms']['uk_ms'], d['fp64']['frac'])"
f=$(find gpurun_out/ab_w$i -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if 'k_ukw<6' in n:
        print(n[:40].ljust(40), r['Calls'], float(r['AverageNs'])/1e3)
PY
done
./build.sh >/dev/null 2>&1
