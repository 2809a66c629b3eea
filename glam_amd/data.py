"""Graph containers, PyG-style collation and synthetic ESOL / protein-shaped graphs.

The reference builds its inputs with RDKit + ``torch_geometric.data`` (neither is
available), so the path's *input layout* is restated here:

* ``Data`` / ``Batch.from_data_list`` follow PyG collation semantics as used by the
  reference ``DataLoader`` (``src_1gp/trainer.py:37-41``): concatenate ``x`` /
  ``edge_attr`` / ``y``, offset ``edge_index`` by the cumulative node count, and emit a
  non-decreasing ``batch`` vector.
* ``synth_molecule`` emits the tensor layout of ``get_mol_nodes_edges``
  (``src_1gp/dataset.py:60-97``): ``x[n,15]`` = one-hot(9) atom type | one-hot(3)
  hybridisation | atomic number, aromatic flag, #H ; ``edge_index`` int64 ``[2,E]`` with
  both directions of every bond, sorted by ``src*n+dst`` (``dataset.py:84-86``);
  ``edge_attr[E,4]`` one-hot bond type.  Sizes follow SURVEY.md §8d: atoms/mol ~
  U{12..28}, chain + ring-closure bonds => E/N ~ 2.05, in-degree 1..4.
* ``synth_protein`` emits the layout of the contact-map graphs of
  ``src_2gi_dti_scr/dataset.py:67-103``: ``x[n,49]``, chain edges then symmetric random
  contacts (unsorted, may hold duplicates), ``edge_attr[E,8]`` continuous.
"""
from __future__ import annotations

import numpy as np
import torch

_ATOMIC_NUMBERS = np.array([1, 6, 7, 8, 9, 16, 17, 35, 53], dtype=np.float32)


def _carry_marks(src, dst):
    """Copy the host-side validation marks of ``src`` onto its device copy ``dst`` — only the marks that are still valid
    for ``src`` (made at its current version counter): a tensor written in place after validation, or a different tensor
    assigned to the field, carries nothing over and is validated on the device like any foreign tensor."""
    if dst is src:
        return
    if getattr(src, "_glam_trusted", None) == src._version:
        dst._glam_trusted = dst._version
    oh = getattr(src, "_glam_onehot", None)
    if oh is not None and oh[1] == src._version:
        dst._glam_onehot = (oh[0], dst._version)


class Data:
    """Minimal attribute bag with the fields the reference model reads
    (``model.py:47-57``: ``x, edge_index, edge_attr, batch``) plus ``y``."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, batch=None, **kw):
        self.x, self.edge_index, self.edge_attr, self.y, self.batch = x, edge_index, edge_attr, y, batch
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        return 0 if self.x is None else self.x.size(0)

    @property
    def num_edges(self):
        return 0 if self.edge_index is None else self.edge_index.size(1)

    def _tensor_items(self):
        return [(k, v) for k, v in self.__dict__.items() if torch.is_tensor(v)]

    def to(self, device, non_blocking=False):
        out = self.__class__.__new__(self.__class__)
        out.__dict__.update(self.__dict__)
        for k, v in self._tensor_items():
            moved = v.to(device, non_blocking=non_blocking)
            _carry_marks(v, moved)
            setattr(out, k, moved)
        # device-side staging caches (CSR) are per-object and per-device
        out.__dict__.pop("_glam_cache", None)
        return out

    def _apply_marks(self):
        """Validation facts established on the host (PackedDataset.collate, on the tensors it has just built) ride on the
        tensors so that the device-side staging needs no read-back: ``edge_index`` / ``batch`` ids valid by construction,
        ``edge_attr`` rows one-hot or not.  Called once, by the collation that built the tensors; ``to()`` never re-derives a
        mark from the field name — it copies the mark of the very tensor it moves, and only while that mark is still valid."""
        for field, (attr, value) in self.__dict__.pop("_glam_marks", {}).items():
            t = getattr(self, field, None)
            if torch.is_tensor(t):      # tied to the tensor's version counter: an in-place write voids the mark
                setattr(t, attr, t._version if attr == "_glam_trusted" else (value, t._version))

    def __repr__(self):
        body = ", ".join(f"{k}={list(v.shape)}" for k, v in self._tensor_items())
        return f"{self.__class__.__name__}({body})"


class Batch(Data):
    """Disjoint union of graphs (PyG ``Batch.from_data_list`` semantics)."""

    num_graphs = 0

    @classmethod
    def from_data_list(cls, data_list):
        """PyG ``Batch.from_data_list`` semantics: concatenate ``x`` / ``edge_attr`` / ``y``, offset ``edge_index`` by the
        running node count, build ``batch``.  Vectorised: one ``repeat_interleave`` each for the offsets and the batch vector
        (a Python loop of per-graph tensor ops cost 40 ms per 1 024 molecules, 40x the device step)."""
        B = len(data_list)
        ns = torch.tensor([d.x.size(0) for d in data_list], dtype=torch.long)
        es = torch.tensor([d.edge_index.size(1) for d in data_list], dtype=torch.long)
        node_off = ns.cumsum(0) - ns
        ei = torch.cat([d.edge_index for d in data_list], 1) if B else torch.zeros(2, 0, dtype=torch.long)
        ei = ei + torch.repeat_interleave(node_off, es, output_size=int(ei.size(1))).unsqueeze(0)
        eas = [d.edge_attr for d in data_list if d.edge_attr is not None]
        ys = [d.y for d in data_list if d.y is not None]
        n_total = int(ns.sum()) if B else 0
        out = cls(x=torch.cat([d.x for d in data_list], 0), edge_index=ei,
                  edge_attr=torch.cat(eas, 0) if eas else None, y=torch.cat(ys, 0) if ys else None,
                  batch=torch.repeat_interleave(torch.arange(B), ns, output_size=n_total))
        out.num_graphs = B
        out.ptr = torch.cat([ns.new_zeros(1), ns.cumsum(0)])
        return out


class _few_threads:
    """Host-side collation is dozens of tiny tensor ops: with torch's default of one intra-op thread per core they spend
    their time waking a 100+ thread pool (6 ms instead of 0.2 ms per 32-molecule batch on a 128-core host).  Same remedy
    as ``torch.utils.data`` workers: a small thread count while collating, restored afterwards."""

    def __init__(self, n=4):
        self.n = n

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev > self.n:
            torch.set_num_threads(self.n)

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)
        return False


class PackedDataset:
    """All graphs of a dataset in flat tensors (node / edge prefix sums), so that collating ANY set of graph ids is a
    handful of vectorised gathers instead of ``len(ids)`` small concatenations (with the device step
    at ~1 ms for 1 024 molecules, host collation is what caps a shuffling loader)."""

    def __init__(self, data_list):
        data_list = list(data_list)
        self.n = len(data_list)
        ns = torch.tensor([d.x.size(0) for d in data_list], dtype=torch.long)
        es = torch.tensor([d.edge_index.size(1) for d in data_list], dtype=torch.long)
        self.ns, self.es = ns, es
        self.node_ptr = torch.cat([ns.new_zeros(1), ns.cumsum(0)])
        self.edge_ptr = torch.cat([es.new_zeros(1), es.cumsum(0)])
        self.x = torch.cat([d.x for d in data_list], 0)
        self.ei = torch.cat([d.edge_index for d in data_list], 1)            # graph-local node ids
        has_ea = [d.edge_attr is not None for d in data_list]
        has_y = [d.y is not None for d in data_list]
        if any(has_ea) != all(has_ea) or any(has_y) != all(has_y):
            raise ValueError("PackedDataset: edge_attr / y must be present for all graphs or for none")
        self.ea = torch.cat([d.edge_attr for d in data_list], 0) if all(has_ea) and data_list else None
        self.y = torch.cat([d.y for d in data_list], 0) if all(has_y) and data_list else None
        self.y_rows = None if self.y is None else torch.tensor([d.y.size(0) for d in data_list], dtype=torch.long)
        # host-side validation, once: what the device-side staging would otherwise read back for every new batch
        loc_n = torch.repeat_interleave(ns, es, output_size=int(self.ei.size(1)))
        self.valid_ids = bool(((self.ei >= 0) & (self.ei < loc_n.unsqueeze(0))).all()) if self.ei.numel() else True
        self.onehot = None if self.ea is None else \
            (bool((((self.ea == 0) | (self.ea == 1)).all() & (self.ea.sum(dim=1) == 1).all())) if self.ea.numel() else True)
        if self.y_rows is not None and not bool((self.y_rows == 1).all()):
            self.y_ptr = torch.cat([self.y_rows.new_zeros(1), self.y_rows.cumsum(0)])
        else:
            self.y_ptr = None

    @staticmethod
    def _ranges(starts, lens, total):
        """Concatenation of ``arange(s, s + l)`` for every (s, l): one repeat_interleave + one arange."""
        off = lens.cumsum(0) - lens
        return torch.repeat_interleave(starts - off, lens, output_size=total) + torch.arange(total)

    def collate(self, ids):
        ids = torch.as_tensor(ids, dtype=torch.long)
        B = int(ids.numel())
        ns, es = self.ns[ids], self.es[ids]
        n_total, e_total = int(ns.sum()), int(es.sum())
        nodes = self._ranges(self.node_ptr[ids], ns, n_total)
        edges = self._ranges(self.edge_ptr[ids], es, e_total)
        node_off = ns.cumsum(0) - ns
        ei = self.ei[:, edges] + torch.repeat_interleave(node_off, es, output_size=e_total).unsqueeze(0)
        if self.y is None:
            y = None
        elif self.y_ptr is None:
            y = self.y[ids]
        else:
            y = self.y[self._ranges(self.y_ptr[ids], self.y_rows[ids], int(self.y_rows[ids].sum()))]
        out = Batch(x=self.x[nodes], edge_index=ei, edge_attr=None if self.ea is None else self.ea[edges], y=y,
                    batch=torch.repeat_interleave(torch.arange(B), ns, output_size=n_total))
        out.num_graphs = B
        out.ptr = torch.cat([ns.new_zeros(1), ns.cumsum(0)])
        out._glam_marks = {"edge_index": ("_glam_trusted", True), "batch": ("_glam_trusted", True)} if self.valid_ids else {}
        if self.onehot is not None:
            out._glam_marks["edge_attr"] = ("_glam_onehot", self.onehot)
        out._apply_marks()
        return out


class DataLoader:
    """Sequential mini-batch iterator over a list of ``Data`` (the reference's train
    loader does not shuffle: ``src_1gp/trainer.py:37-38``).

    ``device`` / ``cache``: without shuffling the batch composition repeats every epoch, so the collated batches
    can be built and moved to the device ONCE and handed out again as the same tensor objects.  The CSR staging of
    ``ops.graph_index`` is keyed on the ``edge_index`` object, so from the second epoch on a step does no
    collation, no host-to-device copy, no CSR build and no validation sync."""

    def __init__(self, dataset, batch_size=32, shuffle=False, seed=0, device=None, cache=None):
        self.dataset, self.batch_size, self.shuffle, self.seed = list(dataset), batch_size, shuffle, seed
        self.device = device
        self.cache = (not shuffle) if cache is None else bool(cache)
        if self.cache and shuffle:
            raise ValueError("DataLoader: cached batches need a fixed order (shuffle=False)")
        self._epoch = 0
        self._batches = None
        self._packed = None

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _collate(self, idx):
        if self._packed is None:
            try:
                self._packed = PackedDataset(self.dataset)
            except (ValueError, AttributeError, RuntimeError):    # heterogeneous records: per-batch concatenation
                self._packed = False
        with _few_threads():
            b = self._packed.collate(idx) if self._packed else Batch.from_data_list([self.dataset[i] for i in idx])
            return b if self.device is None else b.to(self.device)

    def __iter__(self):
        order = np.arange(len(self.dataset))
        if self.shuffle:
            np.random.default_rng(self.seed + self._epoch).shuffle(order)
        self._epoch += 1
        if self.cache:
            if self._batches is None:
                self._batches = [self._collate(order[s:s + self.batch_size]) for s in range(0, len(order), self.batch_size)]
            yield from self._batches
            return
        for s in range(0, len(order), self.batch_size):
            yield self._collate(order[s:s + self.batch_size])


# --------------------------------------------------------------------------------------
# synthetic graphs
# --------------------------------------------------------------------------------------
def _mol_arrays(rng, n_min=12, n_max=28):
    n = int(rng.integers(n_min, n_max + 1))
    bonds = {(i, i + 1) for i in range(n - 1)}
    for _ in range(max(1, n // 10)):
        i = int(rng.integers(0, max(1, n - 3)))
        j = min(n - 1, i + int(rng.integers(3, 6)))
        if j > i:
            bonds.add((i, j))
    bonds = sorted(bonds)
    btype = rng.integers(0, 4, size=len(bonds))
    src = np.array([b[0] for b in bonds] + [b[1] for b in bonds], dtype=np.int64)
    dst = np.array([b[1] for b in bonds] + [b[0] for b in bonds], dtype=np.int64)
    bt = np.concatenate([btype, btype])
    perm = np.argsort(src * n + dst, kind="stable")
    src, dst, bt = src[perm], dst[perm], bt[perm]
    edge_attr = np.zeros((src.size, 4), dtype=np.float32)
    edge_attr[np.arange(src.size), bt] = 1.0
    at = rng.integers(0, 9, size=n)
    hyb = rng.integers(0, 3, size=n)
    x = np.zeros((n, 15), dtype=np.float32)
    x[np.arange(n), at] = 1.0
    x[np.arange(n), 9 + hyb] = 1.0
    x[:, 12] = _ATOMIC_NUMBERS[at]
    x[:, 13] = rng.integers(0, 2, size=n)
    x[:, 14] = rng.integers(0, 4, size=n)
    return x, np.stack([src, dst]), edge_attr


def synth_molecule(rng, n_tasks=1, task="regression"):
    x, ei, ea = _mol_arrays(rng)
    if task == "regression":
        y = rng.standard_normal((1, n_tasks)).astype(np.float32)
    else:  # multi-task labels with -1 = missing (src_1gp/dataset.py:138)
        y = rng.integers(-1, 2, size=(1, n_tasks)).astype(np.float32)
    return Data(torch.from_numpy(x), torch.from_numpy(ei), torch.from_numpy(ea), torch.from_numpy(y))


def synth_protein(rng, n_min=200, n_max=800, contacts_per_res=4.0):
    n = int(rng.integers(n_min, n_max + 1))
    chain = np.arange(n - 1, dtype=np.int64)
    m = int(contacts_per_res * n / 2)
    a = rng.integers(0, n, size=m)
    b = rng.integers(0, n, size=m)
    keep = a != b
    a, b = a[keep], b[keep]
    src = np.concatenate([chain, chain + 1, a, b])
    dst = np.concatenate([chain + 1, chain, b, a])
    x = rng.standard_normal((n, 49)).astype(np.float32)
    ea = rng.random((src.size, 8)).astype(np.float32)
    y = rng.standard_normal((1, 1)).astype(np.float32)
    return Data(torch.from_numpy(x), torch.from_numpy(np.stack([src, dst])), torch.from_numpy(ea),
                torch.from_numpy(y))


def synth_batch(num_graphs, seed=0, n_tasks=1, task="regression"):
    """ESOL-shaped batch: ``num_graphs`` synthetic molecules collated the
    PyG way.  B=1024, seed 0 gives N~20.7k nodes, E~42.5k directed edges."""
    rng = np.random.default_rng(seed)
    xs, eis, eas, bs, ys = [], [], [], [], []
    off = 0
    for g in range(num_graphs):
        x, ei, ea = _mol_arrays(rng)
        xs.append(x)
        eis.append(ei + off)
        eas.append(ea)
        bs.append(np.full(x.shape[0], g, dtype=np.int64))
        off += x.shape[0]
    if task == "regression":
        y = rng.standard_normal((num_graphs, n_tasks)).astype(np.float32)
    else:
        y = rng.integers(-1, 2, size=(num_graphs, n_tasks)).astype(np.float32)
    out = Batch(torch.from_numpy(np.concatenate(xs)), torch.from_numpy(np.concatenate(eis, 1)),
                torch.from_numpy(np.concatenate(eas)), torch.from_numpy(y),
                torch.from_numpy(np.concatenate(bs)))
    out.num_graphs = num_graphs
    sizes = np.array([x.shape[0] for x in xs])
    out.ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]))
    return out


def synth_protein_batch(num_graphs, seed=0, n_min=200, n_max=800):
    rng = np.random.default_rng(seed)
    return Batch.from_data_list([synth_protein(rng, n_min, n_max) for _ in range(num_graphs)])
