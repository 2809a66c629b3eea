#!/bin/bash
# Robustness evidence (GPU box, repo root: bash tools/robust.sh > gpurun_out/TAG_robust.txt): the GPU suite under the A/B switches that change
# the kernels a test launches, every randomised sweep, the leak / misuse probes and smoke().
R=$PWD
run() { echo "== $*"; ( "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-1} ); }
TAIL=1 run python3 -m pytest tests -m gpu -q
GLAM_X3=0 TAIL=1 run python3 -m pytest tests -m gpu -q
# (k_wgrad_x3 at every size — 128 partials per element for k_param_grads — is a fixture of the tests themselves: wgrad_route, layer_wgrad_route)
GLAM_INFER_FWD=0 GLAM_TS_SW=0 TAIL=1 run python3 -m pytest tests -m gpu -q
GLAM_GRU_PRE=0 GLAM_DENSE_SPLITK=0 TAIL=1 run python3 -m pytest tests -m gpu -q
GLAM_GRU_GATES=0 GLAM_NODE_IN_GRU=0 GLAM_RRELU_IN_GEMM=0 GLAM_HEAD_ACT=0 TAIL=1 run python3 -m pytest tests -m gpu -q
for f in tests/sweeps/fuzz_ws.py tests/sweeps/fuzz_parity.py tests/sweeps/fuzz_model.py tests/sweeps/fuzz_dti.py tools/fuzz_gemm.py tools/fuzz_gru.py tools/fuzz_node.py \
         tools/fuzz_dense.py tools/fuzz_graphed.py tools/misuse_probe.py; do
  [ -f $f ] && TAIL=2 run python3 $f
done
TAIL=1 run python3 tools/leak_check.py
TAIL=1 run python3 tools/leak_check.py routed
TAIL=1 run python3 tools/leak_check.py dti
echo "== smoke"; python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
