// Shared pieces of the software-pipelined (ELL-record) aggregate kernels: csrc/triplet_dma.hip (every wave runs the whole
// pipeline) and csrc/triplet_ws.hip (warp-specialised: producer waves run the pipeline, consumer waves the MFMA epilogue).
#pragma once
#include "triplet_kernels.h"

namespace glam {

constexpr int kMetaSlots = 16;           // a_j / edge_attr slots of the one-piece side table (>= 11 packed edges, all written every pass)

struct FwdDmaArgs {
    const float* xw; const float* a_ij; const float* edge_attr; const float* w_edge; const float* M;
    const int* ell_src; const int* ell_eid;      // [N][4] each
    int N; int Cp; float slope;
    float* aggr; float* stats;
    const float* img_upd; const float* bias_p; float* out;     // fused update epilogue (k_triplet_fwd_pipe<..., FUSE = true>, k_triplet_fwd_ws)
};

struct PassMeta { int deg; int off; int tot; int dmax; };   // per lane: its node's degree, packed slot offset; wave-wide edge
                                                            // count and largest degree (wave-uniform)

}  // namespace glam
