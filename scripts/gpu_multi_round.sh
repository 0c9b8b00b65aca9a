mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/pytest_multi.log 2>&1 < /dev/null; tail -25 gpurun_out/pytest_multi.log
python - <<'PY'
import json
for t in ("peer","rccl"):
    l=json.load(open(f"gpurun_out/bench_torchrun_world1_{t}.json"))
    print(t, l["ms_per_step"], l["value"], l["config"]["sharding"], l["collective_check_rel_err"], l["route_sharded"])
PY
