# Run ON THE GPU BOX: extractor parity tests + one-stream stage times of the bench step.  bash tools/quick_stage.sh [tag] [pytest targets...]
T=${1:-q}; shift
mkdir -p gpurun_out/$T
timeout -k 10 300 python -m pytest ${@:-tests/test_gpu_extract.py tests/test_gpu_fuzz.py} -x -q -m gpu > gpurun_out/$T/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/$T/pytest.log
[ $rc -eq 0 ] || exit $rc
python bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed > gpurun_out/$T/s1.json 2>/dev/null || exit 1
python - <<P
import json; d=json.load(open("gpurun_out/$T/s1.json")); print(d["value"], d["roofline"]["stage_ms_per_batch"])
P
