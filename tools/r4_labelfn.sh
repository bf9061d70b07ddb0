#!/bin/bash
# round 4: the new label-function / RCCL tests, then the loader_io leg of the bench
python -m pytest tests/test_imageio.py tests/test_gpu_rccl.py -m gpu -x -q > gpurun_out/r4_labelfn_tests.log 2>&1; tail -5 gpurun_out/r4_labelfn_tests.log
python - > gpurun_out/r4_loader_io.json 2> gpurun_out/r4_loader_io.err <<'PY'
import json, torch, bench
print(json.dumps(bench.loader_io_rate('cuda:0')['end_to_end']))
PY
cat gpurun_out/r4_loader_io.json; tail -3 gpurun_out/r4_loader_io.err
