"""``torch.ops.glam.*`` — the torch-extension front end of the hot path (``csrc/torch_ext.cpp``, built in-tree as
``glam_amd/_glam_torch.so``): functional operators on ``torch.Tensor`` with C++ autograd nodes, validating with
``TORCH_CHECK`` (RuntimeError on dtype / device / contiguity / shape mismatch), launching on the current HIP stream and
forwarding to the C ABI of ``include/glam_hip.h``.  This is the boundary SURVEY.md §8(b) lists:

    csr_from_edge_index(edge_index, N, by=0)      -> (rowptr, nbr, eid, err)
    batch_ptr(batch, num_graphs)                   -> (ptr, err)
    triplet_aggregate(xw, a_ij, edge_attr, w_edge?, M, rowptr, src, eid, colptr, dst, eid_t, heads, slope=0.2) -> aggr
    triplet_layer(x, edge_attr, weight_node, weight_edge, weight_triplet_att, weight_scale, bias, <6 CSR tensors>, heads, slope=0.2) -> out
    segment_pool(x, ptr, mode)  segment_softmax_aggregate(gate, v, ptr)  global_pool5(x, ptr, k=3)  sort_pool_topk_last(x, ptr, k=3)

``load()`` registers the library with torch's dispatcher; there is no fallback: a missing ``.so`` raises."""
from __future__ import annotations

import os

import torch

from . import _lib

EXT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_glam_torch.so")
_loaded = False


def load():
    global _loaded
    if not _loaded:
        if not os.path.exists(EXT_PATH):
            raise _lib.GlamHipError(f"{EXT_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(or `make -C glam_amd/csrc`)")
        _lib.load()                       # libglam_hip.so first: the extension links against it
        torch.ops.load_library(EXT_PATH)
        _loaded = True
    return torch.ops.glam


OPS = ("csr_from_edge_index", "batch_ptr", "triplet_aggregate", "triplet_layer", "segment_pool", "segment_softmax_aggregate",
       "global_pool5", "sort_pool_topk_last")
