#!/bin/bash
# Builds an instrumented copy of libglam_hip.so for the in-kernel cycle profilers of tools/*_prof.py.
# usage: tools/build_prof_variant.sh {b1|b1n|ts|dma|gru|fwd|pg|ws|tl}   ->  tools/tmp/variants/lib_<name>prof.so  (use via GLAM_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../glam_amd/csrc"
make -j8 > /dev/null
case "$1" in
  b1)   src=triplet_h3.hip; def=GLAM_B1_PROF=1 ;;      # stamps drain the queues: where the latency is
  b1n)  src=triplet_h3.hip; def=GLAM_B1_PROF=2 ;;      # stamps do not drain: the overlapped picture
  ts)   src=gemm.hip;       def=GLAM_TS_PROF ;;
  dma)  src=triplet_pipe.hip; def=GLAM_DMA_PROF ;;
  gru)  src=block.hip;       def=GLAM_GRU_PROF ;;
  fwd)  src=triplet_h3.hip;  def=GLAM_FWD_PROF ;;
  pg)   src=layer.hip;       def=GLAM_PG_PROF ;;
  ws)   src=triplet_ws.hip;  def=GLAM_WS_PROF ;;
  tl)   src="triplet_ws.hip triplet_ws_b1.hip block.hip"; def=GLAM_WS_TL ;;   # timeline stamps of the three warp-specialised kernels (tools/ws_timeline.py)
  *) echo "usage: $0 {b1|b1n|ts|dma|gru|fwd|pg|ws|tl}"; exit 2 ;;
esac
mkdir -p ../../tools/tmp/variants
objs=""; keep=$(ls build/*.o)
for f in $src; do
  obj=/tmp/glam_${1}_prof_${f%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -D$def -c $f -o $obj
  objs="$objs $obj"; keep=$(echo "$keep" | grep -v "build/${f%.hip}.o")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/tmp/variants/lib_${1}prof.so $keep $objs
echo "tools/tmp/variants/lib_${1}prof.so"
