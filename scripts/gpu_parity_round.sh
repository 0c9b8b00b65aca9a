mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/pytest_x.log 2>&1 < /dev/null; tail -10 gpurun_out/pytest_x.log
python - <<'PY'
import json
for t in ("peer","rccl"):
    l=json.load(open(f"gpurun_out/bench_torchrun_world1_{t}.json"))
    print(t, l["ms_per_step"], l["config"]["sharding"], l["route_sharded"])
print(open("gpurun_out/dist_world1.json").read())
PY
