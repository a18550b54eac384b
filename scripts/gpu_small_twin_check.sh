#!/bin/bash
# A small user twin on both of its routes: tests, then the two bench lines.   gpurun --timeout 1800 -- 'bash scripts/gpu_small_twin_check.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_user_model.py tests/test_cpp_facade.py -q -m gpu 2>&1 | tail -30
timeout 900 python bench.py --steps 10 --warmup 3 --extra-configs --extras-out gpurun_out/bench_small_extras.json > gpurun_out/bench_small.json 2> gpurun_out/bench_small.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_small_extras.json"))  # (the other configurations live in the extras file since round 6)
for k, v in d["configs"].items():
    if "linear6" in k:
        print(k, "%.3e /s  %.3f ms" % (v["value"], v["ms_per_step"]), {s: round(x["ms_per_launch"], 3) for s, x in v["stages"].items()})
PY
