"""CPU: host-side logic — C-ABI exports, parameter/initialisation/checkpoint compatibility with the
reference (against the golden fixtures), string-dispatched configuration, collation, and loud
failure when no HIP device is behind the tensors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from glam_amd import _lib, layer, model, ops
from glam_amd.data import Batch, Data, DataLoader, synth_batch, synth_molecule, synth_protein_batch
from tests.conftest import ROOT, Golden, assert_close


def _header_functions():
    src = open(os.path.join(ROOT, "include", "glam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(glam_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _header_functions()
    assert len(names) >= 16
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libglam_hip.so does not export {n}"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and include/glam_hip.h disagree"
    assert _lib.load().glam_abi_version() == _lib.ABI_VERSION == 4


def test_abi_rejects_bad_arguments_without_touching_a_gpu():
    lib = _lib.load()
    # Cp not a multiple of 4 -> GLAM_E_INVALID before any launch
    rc = lib.glam_triplet_fwd(None, None, None, None, None, None, None, None, 10, 10, 3, 30, 4, 1, 0.2, None, None, None)
    assert rc == _lib.GLAM_E_INVALID and b"multiple of 4" in lib.glam_last_error()
    rc = lib.glam_triplet_fwd(None, None, None, None, None, None, None, None, 10, 10, 3, 60, 5, 1, 0.2, None, None, None)
    assert rc == _lib.GLAM_E_UNSUPPORTED
    # tensors beyond the 32-bit byte offsets of the aggregate kernels are refused, not silently wrapped
    rc = lib.glam_triplet_fwd(None, None, None, None, None, None, None, None, 6_000_000, 10, 3, 60, 4, 1, 0.2, None, None, None)
    assert rc == _lib.GLAM_E_UNSUPPORTED and b"4 GiB" in lib.glam_last_error()
    rc = lib.glam_pool5_fwd(None, None, 4, 2, 60, 9, None, None, None)
    assert rc == _lib.GLAM_E_UNSUPPORTED
    rc = lib.glam_csr_build(None, 4, 4, 2, None, None, None, None, None, 0, None)
    assert rc == _lib.GLAM_E_INVALID


def test_cpu_tensors_fail_loudly():
    conv = layer.TripletMessage(60, 4)
    b = synth_batch(2, seed=0)
    x = torch.randn(b.x.size(0), 60)
    with pytest.raises(_lib.GlamHipError, match="no CPU fallback"):
        conv(x, b.edge_index, b.edge_attr)
    with pytest.raises(_lib.GlamHipError):
        layer.GlobalPool5()(x, b.batch)
    with pytest.raises(_lib.GlamHipError):
        ops.GraphIndex(b.edge_index, x.size(0))


@pytest.mark.parametrize("name,C,De", [("triplet_c60", 60, 4), ("triplet_c15", 15, 4), ("triplet_de8", 60, 8)])
def test_triplet_init_matches_reference_under_seed(name, C, De):
    """Same parameter creation/initialisation order as layer.py:22-34 => identical weights from the
    same seed (oracle/gen_goldens.py seeds 11 + C + De before constructing the reference layer)."""
    g = Golden(name)
    torch.manual_seed(11 + C + De)
    conv = layer.TripletMessage(C, De)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    sd = conv.state_dict()
    assert list(sd) == ["weight_node", "weight_edge", "weight_triplet_att", "weight_scale", "bias"]
    for k, v in sd.items():
        assert torch.equal(v, g.params[k]), k


def test_light_init_matches_reference_under_seed():
    g = Golden("light_c60")
    torch.manual_seed(31 + 60)
    conv = layer.TripletMessageLight(60, 4)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    for k, v in conv.state_dict().items():
        assert torch.equal(v, g.params[k]), k


@pytest.mark.parametrize("name", ["arch_triplet_pool5", "arch_light_lapool"])
def test_architecture_state_dict_matches_reference(name):
    g = Golden(name)
    m = g.meta
    torch.manual_seed(62)
    np.random.seed(62)
    net = model.Architecture(e_dim=m["e_dim"], out_dim=m["out_dim"], message_steps=m["message_steps"],
                             mol_block=m["mol_block"], mol_readout=m["mol_readout"])
    sd = net.state_dict()
    assert list(sd) == list(g.params), "checkpoint keys / order differ from the reference"
    for k, v in sd.items():
        assert torch.equal(v, g.params[k]), f"{k}: init differs from the reference under the same seed"
    net.load_state_dict(g.params)   # reference checkpoints load unchanged


@pytest.mark.parametrize("name", ["block_triplet_relu", "block_triplet_pair_rrelu", "block_light_celu", "block_nnconv_relu",
                                  "block_gcn_relu", "block_gat_leaky"])
def test_message_block_state_dict_keys(name):
    g = Golden(name)
    m = g.meta
    blk = layer.MessageBlock(60, 60, 4, norm=m["norm"], dropout="_None()", conv=m["conv"], act=m["act"], res=True)
    assert [n for n, _ in blk.named_parameters()] == list(g.params)
    missing = blk.load_state_dict(g.params, strict=False)
    # the goldens store named_parameters(); PyG's GATConv registers the shared Linear twice (lin_l is lin_r)
    assert set(missing.missing_keys) <= {"conv.conv.lin_r.weight"} and not missing.unexpected_keys


def test_string_dispatch_surface():
    for conv in ["_TripletMessage", "_TripletMessageLight", "_NNConv", "_GCNConv", "_GATConv"]:
        blk = layer.MessageBlock(30, 30, 4, norm="_PairNorm", dropout="Dropout(0.1)", conv=conv, act="RReLU", res=1)
        assert (blk.gru is None) == (conv in ["_GCNConv", "_GATConv"])
    for norm in ["_None", "_BatchNorm", "_LayerNorm", "_PairNorm", "_GraphSizeNorm"]:
        for act in ["_None", "ReLU", "LeakyReLU", "RReLU", "CELU", "PReLU"]:
            layer.LinearBlock(8, 4, norm=norm, dropout="_None()", act=act)
    for ro in ["GlobalPool5", "GlobalLAPool", "Set2Set"]:
        net = model.Architecture(mol_block="_TripletMessage", mol_readout=ro, e_dim=32)
        assert net.mol_flat.linear.in_features == (300 if ro == "GlobalPool5" else 120)
    assert layer.TripletMessage(60, 4).extra_repr() == "60, 60, heads=3"
    import inspect
    for cls in (layer.TripletMessage, layer.TripletMessageLight):     # PyG lifts by these names (layer.py:42, :88)
        assert list(inspect.signature(cls.message).parameters)[1:] == ["x_j", "x_i", "edge_index_i", "edge_attr", "size_i"]
        assert list(inspect.signature(cls.update).parameters)[1:] == ["aggr_out"]
        assert callable(cls.propagate)


def test_collation_matches_pyg_semantics():
    rng = np.random.default_rng(3)
    mols = [synth_molecule(rng) for _ in range(5)]
    b = Batch.from_data_list(mols)
    off = 0
    e0 = 0
    for gidx, d in enumerate(mols):
        n, e = d.x.size(0), d.edge_index.size(1)
        assert torch.equal(b.x[off:off + n], d.x)
        assert torch.equal(b.edge_index[:, e0:e0 + e], d.edge_index + off)
        assert torch.equal(b.edge_attr[e0:e0 + e], d.edge_attr)
        assert (b.batch[off:off + n] == gidx).all()
        off += n
        e0 += e
    assert b.num_graphs == 5 and b.y.shape == (5, 1)
    assert (b.batch[1:] >= b.batch[:-1]).all()
    loader = DataLoader(mols, batch_size=2)
    assert len(loader) == 3 and [bb.num_graphs for bb in loader] == [2, 2, 1]


def test_synthetic_batches_have_the_reference_layout():
    b = synth_batch(64, seed=0)
    N, E = b.x.size(0), b.edge_index.size(1)
    assert b.x.shape == (N, 15) and b.edge_attr.shape == (E, 4) and b.edge_index.dtype == torch.int64
    assert 12 * 64 <= N <= 28 * 64 and 1.9 < E / N < 2.3
    src, dst = b.edge_index
    assert (b.batch[src] == b.batch[dst]).all(), "edges must not cross graphs"
    assert (src != dst).all()
    # symmetric: every (s,d) has its (d,s)
    fwd = set(zip(src.tolist(), dst.tolist()))
    assert all((d, s) in fwd for s, d in fwd)
    # per molecule sorted by src*n+dst (dataset.py:84-86) => globally sorted by (src, dst)
    key = src * N + dst
    assert (key[1:] > key[:-1]).all()
    assert torch.equal(b.edge_attr.sum(1), torch.ones(E)) and torch.equal(b.x[:, :9].sum(1), torch.ones(N))
    deg = torch.bincount(dst, minlength=N)
    assert deg.min() >= 1 and deg.max() <= 4
    p = synth_protein_batch(2, seed=1, n_min=50, n_max=80)
    assert p.x.size(1) == 49 and p.edge_attr.size(1) == 8


def test_model_args_filter():
    from types import SimpleNamespace
    args = SimpleNamespace(dataset="esol", seed=1, gpu=0, lr=1e-3, hid_dim_alpha=2, mol_block="_TripletMessage", e_dim=64)
    assert model.model_args(args) == {"hid_dim_alpha": 2, "mol_block": "_TripletMessage", "e_dim": 64}


def test_data_to_drops_device_caches():
    d = Data(torch.zeros(3, 2), torch.zeros(2, 0, dtype=torch.long))
    d._glam_cache = {"x": 1}
    assert not hasattr(d.to("cpu"), "_glam_cache")


def test_flat_view_recognises_bucket_views():
    from glam_amd.parallel import flat_view
    buf = torch.arange(20, dtype=torch.float32)
    a, b, c = (t.view(s) for t, s in zip(buf[2:].split([6, 4, 8]), ((2, 3), (4,), (2, 4))))
    fv = flat_view([a, b, c])
    assert fv is not None and fv.data_ptr() == a.data_ptr() and fv.numel() == 18
    fv.mul_(2)
    assert float(c[1, 3]) == 38.0                       # reductions on the bucket show through the gradients
    assert flat_view([a, c]) is None                    # gap
    assert flat_view([a, torch.zeros(4)]) is None       # different storage
    assert flat_view([]) is None


def test_dataloader_caches_collated_batches_when_order_is_fixed():
    rng = np.random.default_rng(5)
    mols = [synth_molecule(rng) for _ in range(7)]
    loader = DataLoader(mols, batch_size=3)                  # shuffle=False => cached
    first = list(loader)
    second = list(loader)
    assert len(first) == 3 and all(a is b for a, b in zip(first, second))       # same objects: CSR cache keys stay valid
    assert all(a.edge_index is b.edge_index for a, b in zip(first, second))
    fresh = list(DataLoader(mols, batch_size=3, cache=False))
    assert all(torch.equal(a.x, b.x) and torch.equal(a.edge_index, b.edge_index) for a, b in zip(first, fresh))
    shuffled = DataLoader(mols, batch_size=3, shuffle=True, seed=1)
    assert [b.num_graphs for b in shuffled] == [3, 3, 1] and shuffled.cache is False
    with pytest.raises(ValueError):
        DataLoader(mols, batch_size=3, shuffle=True, cache=True)


def test_gpu_manager_surface(monkeypatch):
    """Search-loop device picker (utils.py:185-246) without nvidia-smi: same keys and choices, fed by the HIP runtime."""
    from glam_amd.devices import GPUManager
    if not torch.cuda.is_available():
        m = GPUManager()
        assert m.gpu_num == 0 and m.auto_choice(0.5) is None and m.wait_free_gpu() == -1
    mem = {0: (10 << 30, 288 << 30), 1: (250 << 30, 288 << 30), 2: (200 << 30, 288 << 30)}
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 3)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda i: mem[i])
    monkeypatch.setattr(torch.cuda, "get_device_name", lambda i: "AMD Instinct MI355X")
    m = GPUManager(["power.draw"])
    assert m.gpu_num == 3 and set(m.gpus[0]) == {"index", "gpu_name", "memory.free", "memory.total", "power.draw"}
    assert m.gpus[1]["memory.free"] == 250 * 1024 and m.auto_choice(0.7) == 1 and m.wait_free_gpu(0.7) == 1
    mem[1] = (100 << 30, 288 << 30)
    assert m.auto_choice(0.7) is None and m.auto_choice(0.5) == 2


def test_packed_dataset_collates_like_from_data_list():
    """Vectorised collation (flat dataset + gathers) == PyG-style concatenation, for contiguous and shuffled graph ids,
    single-row and multi-row labels, graphs without edges; the DataLoader uses it for every batch."""
    import numpy as np
    from glam_amd.data import Batch, Data, DataLoader, PackedDataset, synth_molecule
    rng = np.random.default_rng(3)
    mols = [synth_molecule(rng) for _ in range(40)]
    mols[7] = Data(mols[7].x[:1], torch.zeros(2, 0, dtype=torch.long), torch.zeros(0, 4), y=mols[7].y)     # single atom, no bonds
    pk = PackedDataset(mols)
    for ids in (np.arange(0, 16), np.arange(30, 40), rng.permutation(40)[:13], np.array([7]), np.array([], dtype=np.int64)):
        a = Batch.from_data_list([mols[i] for i in ids]) if len(ids) else None
        b = pk.collate(ids)
        if a is None:
            assert b.x.size(0) == 0 and b.edge_index.size(1) == 0 and b.num_graphs == 0
            continue
        for k in ("x", "edge_index", "edge_attr", "y", "batch", "ptr"):
            assert torch.equal(getattr(a, k), getattr(b, k)), k
        assert a.num_graphs == b.num_graphs
    for m in mols:                                              # two label rows per graph
        m.y = torch.randn(2, 3)
    pk2 = PackedDataset(mols)
    ids = rng.permutation(40)[:9]
    assert torch.equal(pk2.collate(ids).y, Batch.from_data_list([mols[i] for i in ids]).y)
    got = [b.x.size(0) for b in DataLoader(mols, batch_size=16, shuffle=True, seed=5)]
    assert sum(got) == sum(m.x.size(0) for m in mols) and len(got) == 3


def test_loader_batches_carry_host_side_validation_marks():
    """PackedDataset validates ids / one-hot rows once on the host; the marks ride on the collated tensors (and survive
    ``.to()``), so the device-side staging reads nothing back for loader-built batches."""
    import numpy as np
    from glam_amd.data import DataLoader, PackedDataset, synth_molecule
    rng = np.random.default_rng(1)
    mols = [synth_molecule(rng) for _ in range(10)]
    b = next(iter(DataLoader(mols, batch_size=4, shuffle=True, seed=2)))
    b2 = b.to("cpu")
    for t in (b.edge_index, b.batch, b2.edge_index, b2.batch):
        assert getattr(t, "_glam_trusted", None) == t._version
    assert b2.edge_attr._glam_onehot == (True, b2.edge_attr._version)
    b2.edge_index[0, 0] = 0                                            # an in-place write voids the mark
    assert b2.edge_index._glam_trusted != b2.edge_index._version
    mols[3].edge_attr = mols[3].edge_attr * 0.5                       # not one-hot any more
    assert PackedDataset(mols).collate(np.arange(4)).edge_attr._glam_onehot[0] is False
    mols[5].edge_index = mols[5].edge_index.clone()
    mols[5].edge_index[0, 0] = 99                                      # an id outside its graph: no trust mark, the device checks
    bad = PackedDataset(mols).collate(np.arange(4, 8))
    assert getattr(bad.edge_index, "_glam_trusted", None) is None


def test_validation_marks_do_not_survive_a_write_or_a_replacement_across_to():
    """``Data.to()`` copies the mark of the very tensor it moves, and only while that mark is valid: a tensor written in
    place after collation, or a different tensor assigned to the field, arrives on the device unmarked (so the device-side
    staging validates it: R-GCN one-hot path not taken on scaled features, bad ids raise IndexError)."""
    import numpy as np
    from glam_amd.data import PackedDataset, synth_molecule
    rng = np.random.default_rng(4)
    mols = [synth_molecule(rng) for _ in range(6)]
    pk = PackedDataset(mols)
    meta = torch.device("meta")                                        # a real copy (new tensor objects) without a GPU

    b = pk.collate(np.arange(6))
    ok = b.to(meta)
    assert ok.edge_index._glam_trusted == ok.edge_index._version and ok.batch._glam_trusted == ok.batch._version
    assert ok.edge_attr._glam_onehot == (True, ok.edge_attr._version)

    b = pk.collate(np.arange(6))
    b.edge_attr.mul_(0.5)                                              # write, then to(): no one-hot mark on the copy
    moved = b.to(meta)
    assert getattr(moved.edge_attr, "_glam_onehot", None) is None
    assert moved.edge_index._glam_trusted == moved.edge_index._version   # untouched fields keep theirs

    b = pk.collate(np.arange(6))
    b.edge_index = torch.full((2, 3), 10 ** 6)                         # replace, then to(): the new tensor was never validated
    b.batch = b.batch.clone()
    moved = b.to(meta)
    assert getattr(moved.edge_index, "_glam_trusted", None) is None
    assert getattr(moved.batch, "_glam_trusted", None) is None


def test_torch_extension_registers_the_boundary_ops_and_refuses_cpu_tensors():
    """the hot path is exposed through a torch extension (TORCH_LIBRARY(glam)): every operator of the list is
    registered with torch's dispatcher from the in-tree _glam_torch.so, and a CPU tensor is a RuntimeError (TORCH_CHECK), never a
    silent CPU computation."""
    from glam_amd import torch_ext
    ns = torch_ext.load()
    for name in torch_ext.OPS:
        assert hasattr(ns, name), name
    sch = str(ns.triplet_layer.default._schema)
    assert "weight_triplet_att" in sch and "int heads" in sch
    with pytest.raises(RuntimeError, match="HIP device only"):
        ns.csr_from_edge_index(torch.zeros(2, 3, dtype=torch.long), 4, 0)
    with pytest.raises(RuntimeError, match="HIP device only"):
        ns.segment_pool(torch.zeros(4, 8), torch.zeros(3, dtype=torch.int32), 0)


def test_adam_restatement_matches_the_library_optimizer_on_cpu():
    """The arithmetic of ``csrc/optim.hip`` (``oracle.adam_step``: fp32, bias corrections as -expm1(s ln beta)) against
    ``torch.optim.Adam`` — the optimizer the reference's trainer constructs (trainer.py:49-50) — over 60 steps with a learning-rate
    change and weight decay: 2e-6 of the largest parameter, moments to 1e-6.  The GPU test compares the kernel with the same reference."""
    from oracle import glam_oracle as oracle
    rng = np.random.default_rng(5)
    for wd in (0.0, 1e-2):
        p0 = rng.standard_normal(257).astype(np.float32)
        tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
        opt = torch.optim.Adam([tp], lr=1e-2, weight_decay=wd)
        p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
        lr = 1e-2
        for step in range(60):
            if step == 20:
                lr = opt.param_groups[0]["lr"] = 3e-3
            g = (rng.standard_normal(257) * 10.0 ** rng.integers(-3, 2)).astype(np.float32)
            tp.grad = torch.from_numpy(g.copy())
            opt.step()
            p, m, v = oracle.adam_step(p, g, m, v, step, lr=lr, weight_decay=wd)
        scale = max(1.0, float(np.abs(p).max()))
        assert np.abs(p - tp.detach().numpy()).max() <= 2e-6 * scale
        st = opt.state[tp]
        assert np.abs(m - st["exp_avg"].numpy()).max() <= 1e-6 * max(1.0, float(np.abs(m).max()))
        assert np.abs(v - st["exp_avg_sq"].numpy()).max() <= 1e-6 * max(1.0, float(np.abs(v).max()))


def test_adam_abi_rejects_bad_arguments_without_touching_a_gpu():
    import ctypes
    lib = _lib.load()
    assert lib.glam_adam_max_tensors() >= 36          # a default-shaped model's parameter list goes in one launch
    tab = (ctypes.c_uint64 * 4)(0, 0, 0, 0)
    numel = (ctypes.c_int64 * 1)(8)
    args = (0.001, 0.9, 0.999, 1e-8, 0.0, None)
    assert lib.glam_adam_step(tab, numel, 1, None, None, None, *args) == _lib.GLAM_E_INVALID            # no step counter / ticket
    assert lib.glam_adam_step(tab, numel, 1, 8, 8, None, *args) == _lib.GLAM_E_INVALID                  # null tensor addresses
    assert lib.glam_adam_step(tab, numel, 1, 8, 8, None, 0.001, 1.0, 0.999, 1e-8, 0.0, None) == _lib.GLAM_E_INVALID   # beta1 = 1
    assert lib.glam_adam_step(tab, numel, 0, 8, 8, None, *args) == 0                                    # nothing to do


def test_padded_view_registry_is_host_logic():
    """ops.slice_cols / pad_cols / padded_base (the zero-padded flow of the odd hidden widths): pure bookkeeping, no kernels.  A view
    finds its padded tensor by address, strides and version; a view of a VIEW keeps the registered object alive; an in-place write or a
    different stride ends the trust."""
    import gc
    xp = torch.zeros(6, 48)
    xp[:, :45] = torch.randn(6, 45)
    v = ops.slice_cols(xp, 45)
    assert v.shape == (6, 45) and ops.padded_base(v) is xp and ops.pad_cols(v, 48) is xp
    assert ops.padded_base(v.unsqueeze(0).squeeze(0)) is xp                 # any view with the same address and strides
    assert ops.padded_base(xp) is None and ops.slice_cols(xp, 48) is xp     # nothing to slice
    # a registered tensor that is itself a view (the skip-connection alias of the layer node): its slice keeps it alive
    alias = xp.view_as(xp)
    w = ops.slice_cols(alias, 45)
    del alias
    gc.collect()
    assert ops.padded_base(w) is not None and ops.padded_base(w).data_ptr() == xp.data_ptr()
    w.add_(1.0)                                                             # the pad columns may no longer be what the producer left
    assert ops.padded_base(w) is None
    p = ops.pad_cols(w, 48)
    assert p.shape == (6, 48) and torch.equal(p[:, :45], w) and (p[:, 45:] == 0).all()
    other = torch.zeros(6, 52)[:, :45]                                      # a view nobody registered, another pitch
    assert ops.padded_base(other) is None


def test_pad_group_and_new_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument checks of the entry points added in round 4 run before any device work."""
    import ctypes
    lib = _lib.load()
    one = (ctypes.c_void_p * 1)(1 << 20)
    assert lib.glam_pad_group(9, one, one, (ctypes.c_int32 * 45)(), 0, None) == _lib.GLAM_E_INVALID
    assert lib.glam_pad_group(1, one, one, (ctypes.c_int32 * 5)(1, 4, 4, 3, 4), 0, None) == _lib.GLAM_E_INVALID     # padded < plain
    assert lib.glam_pad_group(0, None, None, None, 0, None) == 0
    assert lib.glam_gru_ws_supported(60) == 1 and lib.glam_gru_ws_supported(20) == 0 and lib.glam_gru_ws_supported(62) == 0
    assert lib.glam_gru_ws_fwd(None, None, None, None, None, None, None, 4, 20, 0, 1, 0.0, None, None, None, None, None) == _lib.GLAM_E_UNSUPPORTED
    assert lib.glam_gru_bwd_ws(None, None, None, None, None, None, None, None, None, 4, 62, 0, 1, 0.0, 0, None, None, None, None, None, None) == _lib.GLAM_E_UNSUPPORTED
    assert lib.glam_pool5_padded_fwd(None, None, 4, 2, 50, 45, 3, None, None, None) == _lib.GLAM_E_INVALID             # ld is not ceil4(D)
    assert lib.glam_ts_gemm_image_bytes(276, 92) > 0 and lib.glam_ts_gemm_image_bytes(300, 92) >= 0


def test_dense_gemm_entry_points_reject_bad_arguments_without_a_gpu():
    """glam_dense_gemm / glam_linear_dense_fwd / _bwd check strides, sizes and the all-ones column's layout before any launch."""
    lib = _lib.load()
    p = 1 << 20
    inv = _lib.GLAM_E_INVALID
    assert lib.glam_dense_gemm(p, 300, 2, None, 0.0, p, 1, 300, None, 0, 0.0, p, 64, None, 8, 64, 300, None) == inv      # no unit stride in A
    assert b"A's strides" in lib.glam_last_error()
    assert lib.glam_dense_gemm(p, 300, 1, None, 0.0, p, 3, 300, None, 0, 0.0, p, 64, None, 8, 64, 300, None) == inv      # ... nor in B
    assert lib.glam_dense_gemm(p, 300, 1, None, 0.0, p, 1, 300, None, 0, 0.0, p, 64, p, 8, 64, 300, None) == inv         # ones column, B along k
    assert lib.glam_dense_gemm(p, 300, 1, None, 0.0, p, 1, 300, None, 3, 0.0, p, 64, None, 8, 64, 300, None) == inv      # unknown activation
    assert lib.glam_dense_gemm(p, 300, 1, None, 0.0, p, 1, 300, None, 0, 0.0, p, 60, None, 8, 64, 300, None) == inv      # ldc < Cn
    assert lib.glam_dense_gemm(p, 2, 1, None, 0.0, p, 1, 2, None, 0, 0.0, p, 64, None, 8, 64, 2, None) == inv            # K < 4
    assert lib.glam_dense_gemm(None, 300, 1, None, 0.0, p, 1, 300, None, 0, 0.0, p, 64, None, 8, 64, 300, None) == inv   # null operand
    assert lib.glam_linear_dense_fwd(p, p, None, 0, 300, 64, 0, 0.0, p, None) == inv
    assert lib.glam_linear_dense_bwd(p, p, p, None, 0.0, 2, 300, 64, p, p, None, None) == inv                            # N < 4
    assert lib.glam_linear_dense_bwd(p, p, p, None, 0.0, 8, 300, 64, None, None, None, None) == inv                      # nothing to compute
    assert lib.glam_linear_dense_bwd(p, p, p, None, 0.0, 8, 300, 64, p, None, p, None) == inv                            # db without dw
    assert ops.linear_dense_supported(1024, 300, 1024) and not ops.linear_dense_supported(1024, 450, 1024) and not ops.linear_dense_supported(64, 1024, 617)
    assert not ops.linear_dense_supported(2, 300, 1024) and not ops.linear_dense_supported(1024, 16, 1024)
    assert lib.glam_prestage(None, None, None, None, None, 0, 0, 0, 0, 0, None, 0, None, None, None, None) == inv          # nothing to build
    assert lib.glam_prestage(None, None, None, None, None, 0, 0, 0, 0, 0, None, 7, None, None, None, None) == inv          # more than six images


def test_launch_saving_hints_are_host_logic():
    """following_dropout (the hint that lets an activation launch write the dropped twin), the prestage item lists and the operand-set
    thresholds are decided on the host: a training-mode Dropout with nothing in front of it gives its p, anything else 0; CPU tensors
    stage nothing."""
    blk = layer.LinearBlock(1024, 1, norm="_None", dropout="Dropout(0.2)", act="_None")
    assert layer.following_dropout(blk.train()) == pytest.approx(0.2)
    assert layer.following_dropout(blk.eval()) == 0.0
    assert layer.following_dropout(layer.LinearBlock(60, 60, norm="_PairNorm", dropout="Dropout(0.2)").train()) == 0.0     # a norm sits in front
    assert layer.following_dropout(layer.LinearBlock(60, 60, norm="_None", dropout="_None()").train()) == 0.0
    mb = layer.MessageBlock(60, 60, 4, norm="_None", dropout="Dropout(0.1)", conv="_TripletMessage", act="RReLU()").train()
    assert layer.following_dropout(mb) == pytest.approx(0.1)
    lin = layer.LinearBlock(15, 60)
    x, ea = torch.randn(40, 15), torch.zeros(80, 4)
    assert layer.prestage_pass((lin, mb, x, ea)) == 0                 # CPU tensors: nothing to stage, nothing launched
    trip, images, pres = layer._prestage_items(lin, mb, x.to("meta") if False else x, ea)
    assert trip is None and images == [] and pres == []
    assert ops.prestage(None, []) == 0 and ops.prestage(None, [], [(torch.zeros(180, 60), torch.zeros(180, 60), 60)]) == 0      # no scope: nothing built
    # ops.cat_cols outside its class (CPU tensors, no gradient wanted, more than eight pieces) is torch.cat
    a, b = torch.randn(3, 2, requires_grad=True), torch.randn(3, 4)
    assert torch.equal(ops.cat_cols([a, b]), torch.cat([a, b], dim=-1)) and not type(ops.cat_cols([a, b]).grad_fn).__name__.startswith("_CatCols")
    assert ops.cat_cols([torch.zeros(2, 1)] * 9).shape == (2, 9)
    assert ops.GRU_WGRAD_BATCH in (True, False) and ops.PRESTAGE in (True, False) and ops.NORM_DROP in (True, False) and ops.DENSE_LINEAR in (True, False)

