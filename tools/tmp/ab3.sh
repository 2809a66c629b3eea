for i in 1 2; do
for lib in glam_amd/variants/lib_before.so glam_amd/libglam_hip.so; do
  echo "== $lib"
  for preset in relu run_default model_default; do GLAM_HIP_LIB=$PWD/$lib python3 tools/bench_model.py --preset $preset 2>/dev/null | tail -1 | cut -c1-60,150-330; done
done; done
bash tools/tmp/pm.sh
