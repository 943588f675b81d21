python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t.log 2>&1; tail -4 gpurun_out/t.log; grep -E "^E  " gpurun_out/t.log | head -6
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench.log") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], "fast:", d.get("fast_math") and d["fast_math"]["value"])
for k,v in list(d["kernel_ms_per_step"].items())[:12]: print(k, round(v["launches_per_step"]), round(v["ms_per_step"],3))
PY
