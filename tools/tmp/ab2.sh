for i in 1 2; do
for lib in glam_amd/variants/lib_before.so glam_amd/libglam_hip.so; do
  echo "== $lib"
  for preset in relu run_default; do GLAM_HIP_LIB=$PWD/$lib python3 tools/bench_model.py --preset $preset 2>/dev/null | tail -1; done
  GLAM_HIP_LIB=$PWD/$lib python3 tools/bench_model.py --preset relu --batch 32 2>/dev/null | tail -1
done; done
GLAM_HIP_LIB=$PWD/glam_amd/libglam_hip.so python3 tools/bench_dti.py 2>/dev/null | tail -3
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
