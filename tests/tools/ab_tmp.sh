for v in "-DTWX_DT_WAVES=8" "-DTWX_DT_WAVES=4"; do
./build.sh $v >/dev/null 2>&1
python bench.py --steps 4 --warmup 1 --daily-years 69 --stream-tiles 0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b69.json
python - <<PY
import json
d=json.load(open('gpurun_out/b69.json'))['daily']; print("$v", d['ms_per_step'], d['timing_ms']['daily_ms'])
PY
done
./build.sh -DTWX_DT_WAVES=8 >/dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
./build.sh >/dev/null 2>&1
