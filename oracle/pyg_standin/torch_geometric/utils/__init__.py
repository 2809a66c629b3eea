"""Stand-in for ``torch_geometric.utils`` (test infrastructure; see package docstring)."""
import torch


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    """``torch_scatter.scatter`` along ``dim=0`` for 1-D ``index`` (rows of ``src``).

    Empty segments yield 0 for every reduction (torch-scatter convention)."""
    assert dim in (0, -src.dim()), "stand-in scatters along dim 0 only"
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    out_shape = (dim_size,) + tuple(src.shape[1:])
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    if reduce in ("sum", "add"):
        return src.new_zeros(out_shape).scatter_add_(0, idx, src)
    if reduce == "mean":
        tot = src.new_zeros(out_shape).scatter_add_(0, idx, src)
        cnt = src.new_zeros(dim_size).scatter_add_(0, index, src.new_ones(index.shape))
        cnt = cnt.clamp_(min=1).view((-1,) + (1,) * (src.dim() - 1))
        return tot / cnt
    if reduce == "max":
        out = src.new_zeros(out_shape)
        return out.scatter_reduce(0, idx, src, reduce="amax", include_self=False)
    if reduce == "min":
        out = src.new_zeros(out_shape)
        return out.scatter_reduce(0, idx, src, reduce="amin", include_self=False)
    raise ValueError(reduce)


def softmax(src, index, ptr=None, num_nodes=None):
    """PyG 1.7.2 ``utils.softmax``: max-shifted exp grouped by ``index`` with a
    ``+1e-16`` in the denominator; each trailing column independently."""
    if num_nodes is None:
        num_nodes = int(index.max()) + 1 if index.numel() > 0 else 0
    src_max = scatter(src, index, 0, num_nodes, "max")
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = scatter(out, index, 0, num_nodes, "sum").index_select(0, index)
    return out / (out_sum + 1e-16)


def degree(index, num_nodes=None, dtype=None):
    if num_nodes is None:
        num_nodes = int(index.max()) + 1 if index.numel() > 0 else 0
    out = torch.zeros(num_nodes, dtype=dtype or torch.get_default_dtype(), device=index.device)
    return out.scatter_add_(0, index, out.new_ones(index.shape))


def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    return edge_index, (None if edge_attr is None else edge_attr[mask])


def add_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    loop = loop.unsqueeze(0).repeat(2, 1)
    if edge_weight is not None:
        edge_weight = torch.cat([edge_weight, edge_weight.new_full((num_nodes,), fill_value)])
    return torch.cat([edge_index, loop], dim=1), edge_weight


def add_remaining_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    row, col = edge_index
    mask = row != col
    loop_weight = None
    if edge_weight is not None:
        loop_weight = edge_weight.new_full((num_nodes,), fill_value)
        inv = ~mask
        loop_weight[row[inv]] = edge_weight[inv]
        edge_weight = torch.cat([edge_weight[mask], loop_weight])
    loop = torch.arange(num_nodes, dtype=row.dtype, device=row.device).unsqueeze(0).repeat(2, 1)
    return torch.cat([edge_index[:, mask], loop], dim=1), edge_weight


def to_dense_batch(x, batch, fill_value=0.0):
    B = int(batch.max()) + 1
    num = degree(batch, B, dtype=torch.long)
    cum = torch.cat([num.new_zeros(1), num.cumsum(0)])
    nmax = int(num.max())
    pos = torch.arange(batch.numel(), device=x.device) - cum[batch]
    dense = x.new_full((B, nmax) + tuple(x.shape[1:]), fill_value)
    dense[batch, pos] = x
    mask = torch.zeros(B, nmax, dtype=torch.bool, device=x.device)
    mask[batch, pos] = True
    return dense, mask
