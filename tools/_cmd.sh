R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -1
for v in _base "" _base ""; do
  MSPL_HIP_LIB=$R/mspl_amd/lib/libmspl_hip$v.so python - <<PY
import sys; sys.path.insert(0, "$R")
import bench
r = bench.cityscapes_rate('cuda:0') if hasattr(bench, 'cityscapes_rate') else None
print("lib$v", r and r.get('value'), r and r.get('single_in_flight', {}).get('value') if isinstance(r, dict) else None)
PY
done
