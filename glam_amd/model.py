"""MI355X drop-in for the reference's model assembly (``src_1gp/model.py``).

``Architecture`` / ``Model`` keep the reference constructor keywords (``model.py:24-33``), the
sub-module names (checkpoint keys ``mol_lin0.*``, ``mol_conv.*``, ``mol_readout.*``, ``mol_flat.*``,
``lin_out1.*``) and the call ``model(batch) -> [B, out_dim]`` used by the reference trainers
(``trainer.py:295``).  ``ArchitectureDDI`` mirrors the two-drug variant (``src_2gi_ddi/model.py:9-62``), ``ArchitectureDTI`` the two-tower variant
(``src_2gi_dti_scr/model.py:14-68``).
"""
from __future__ import annotations

import torch

from . import graphs, ops

from .layer import _None  # noqa: F401
from .layer import GlobalPool5, GlobalLAPool, Set2Set  # noqa: F401  (resolved from config strings)
from .layer import LinearBlock, MessageBlock, dot_and_global_pool2, first_node_spec, flat_then_head, following_dropout, prestage_pass


def model_args(args):
    """Filter a run.py-style ``Namespace`` down to constructor keywords (model.py:7-15)."""
    other = ["dataset_root", "dataset", "split", "seed", "gpu", "note", "batch_size", "epochs", "loss", "optim", "k",
             "lr", "lr_reduce_rate", "lr_reduce_patience", "early_stop_patience", "verbose_patience", "split_seed",
             "test"]
    return {k: v for k, v in args.__dict__.items() if k not in other}


def _readout(name, hid_dim):
    return eval("{}(in_channels=hid_dim, processing_steps=3)".format(name), globals(), {"hid_dim": hid_dim})  # noqa: S307


class Architecture(torch.nn.Module):
    def __init__(self, mol_in_dim=15, mol_edge_in_dim=4, hid_dim_alpha=4, e_dim=1024, out_dim=1,
                 mol_block="_NNConv", message_steps=3, mol_readout="GlobalPool5",
                 pre_norm="_None", graph_norm="_None", flat_norm="_None", end_norm="_None",
                 pre_do="_None()", graph_do="Dropout(0.2)", flat_do="_None()", end_do="Dropout(0.2)",
                 pre_act="RReLU", graph_act="RReLU", flat_act="RReLU", graph_res=True):
        super().__init__()
        hid_dim = mol_in_dim * hid_dim_alpha
        self.mol_lin0 = LinearBlock(mol_in_dim, hid_dim, norm=pre_norm, dropout=pre_do, act=pre_act)
        self.mol_conv = MessageBlock(hid_dim, hid_dim, mol_edge_in_dim, norm=graph_norm, dropout=graph_do,
                                     conv=mol_block, act=graph_act, res=graph_res)
        self.message_steps = message_steps
        self.mol_readout = _readout(mol_readout, hid_dim)
        _mol_ro = 5 if mol_readout == "GlobalPool5" else 2
        self.mol_flat = LinearBlock(_mol_ro * hid_dim, e_dim, norm=flat_norm, dropout=flat_do, act=flat_act)
        self.lin_out1 = LinearBlock(e_dim, out_dim, norm=end_norm, dropout=end_do, act="_None")

    def forward(self, data_mol):
        # a batch content seen before is replayed from hipGraphs (glam_amd.graphs.GraphedCallable): the reference's eager training
        # loop (src_1gp/trainer.py:286-304) runs unchanged at graphed speed; ``model.graphed_call = False`` keeps every call eager
        return graphs.graphed_call(self, self._eager_forward, data_mol)

    def _eager_forward(self, data_mol):
        with ops.weight_scope():     # weight re-layouts are shared by the message_steps applications of the block
            out = self._forward(data_mol)
        ops.poll_checks()            # deferred id checks of foreign batches (GLAM_VALIDATE=deferred) whose flag has come back
        return out

    def _forward(self, data_mol):
        prestage_pass((self.mol_lin0, self.mol_conv, data_mol.x, data_mol.edge_attr))  # (the pass's weight re-layouts from one launch)
        # (next_dropout: the block behind applies Dropout to this output first — the activation's launch writes the dropped twin)
        # (... and next_node: that block's TripletMessage reads this output — the embedding's launch writes its node product too)
        xm = self.mol_lin0(data_mol.x, batch=data_mol.batch, next_dropout=following_dropout(self.mol_conv),     # model.py:49
                           next_node=first_node_spec(self.mol_conv, data_mol.x.size(0), data_mol.edge_index, data_mol.edge_attr))
        hm = None
        for i in range(self.message_steps):                                        # model.py:53-54
            with ops.block_feeds_itself(i + 1 < self.message_steps):      # (its output is the same block's next input)
                xm, hm = self.mol_conv(xm, data_mol.edge_index, data_mol.edge_attr, h=hm, batch=data_mol.batch)
        # PyG's pools read the graph count back from ``batch``; a collated Batch already knows it
        num_graphs = getattr(data_mol, "num_graphs", None) or None
        outm = self.mol_readout(xm, data_mol.batch, num_graphs)                    # model.py:57
        # model.py:60-61 — in the default configuration in training mode the head applies mol_flat's RReLU and its own Dropout to
        # the elements it reads (layer.flat_then_head)
        return flat_then_head(self.mol_flat, self.lin_out1, outm)


Model = Architecture


class ArchitectureDTI(torch.nn.Module):
    """Ligand + protein two-tower model with per-pair fusion (src_2gi_dti_scr/model.py:14-68)."""

    def __init__(self, mol_in_dim=15, pro_in_dim=49, mol_edge_in_dim=4, pro_edge_in_dim=8, hid_dim_alpha=4,
                 e_dim=1024, out_dim=1, mol_block="_NNConv", pro_block="_GCNConv", message_steps=3,
                 mol_readout="GlobalPool5", pro_readout="GlobalPool5",
                 pre_norm="_None", graph_norm="_None", flat_norm="_None", end_norm="_None",
                 pre_do="_None()", graph_do="Dropout(0.2)", flat_do="_None()", end_do="Dropout(0.2)",
                 pre_act="RReLU", graph_act="RReLU", flat_act="RReLU", end_act="RReLU", graph_res=True):
        super().__init__()
        hid_dim = mol_in_dim * hid_dim_alpha
        self.mol_lin0 = LinearBlock(mol_in_dim, hid_dim, norm=pre_norm, dropout=pre_do, act=pre_act)
        self.pro_lin0 = LinearBlock(pro_in_dim, hid_dim, norm=pre_norm, dropout=pre_do, act=pre_act)
        self.mol_conv = MessageBlock(hid_dim, hid_dim, mol_edge_in_dim, norm=graph_norm, dropout=graph_do,
                                     conv=mol_block, act=graph_act, res=graph_res)
        self.pro_conv = MessageBlock(hid_dim, hid_dim, pro_edge_in_dim, norm=graph_norm, dropout=graph_do,
                                     conv=pro_block, act=graph_act, res=graph_res)
        self.message_steps = message_steps
        self.mol_readout = _readout(mol_readout, hid_dim)
        self.pro_readout = _readout(pro_readout, hid_dim)
        _mol_ro = 5 if mol_readout == "GlobalPool5" else 2
        _pro_ro = 5 if pro_readout == "GlobalPool5" else 2
        self.mol_flat = LinearBlock(_mol_ro * hid_dim, hid_dim, norm=flat_norm, dropout=flat_do, act=flat_act)
        self.pro_flat = LinearBlock(_pro_ro * hid_dim, hid_dim, norm=flat_norm, dropout=flat_do, act=flat_act)
        self.lin_out0 = LinearBlock(hid_dim * 2 + message_steps * 2, e_dim, norm=end_norm, dropout=end_do, act=end_act)
        self.lin_out1 = LinearBlock(e_dim, out_dim, norm=end_norm, dropout=end_do, act="_None")

    def forward(self, data_mol, data_pro):
        return graphs.graphed_call(self, self._eager_forward, data_mol, data_pro)      # (see Architecture.forward)

    def _eager_forward(self, data_mol, data_pro):
        with ops.weight_scope():
            return self._forward(data_mol, data_pro)

    def _forward(self, data_mol, data_pro):
        prestage_pass((self.mol_lin0, self.mol_conv, data_mol.x, data_mol.edge_attr),
                      (self.pro_lin0, self.pro_conv, data_pro.x, data_pro.edge_attr))
        xm = self.mol_lin0(data_mol.x, batch=data_mol.batch)
        xp = self.pro_lin0(data_pro.x, batch=data_pro.batch)
        hm, hp = None, None
        fusion = []
        for i in range(self.message_steps):
            with ops.block_feeds_itself(i + 1 < self.message_steps):      # (its output is the same block's next input)
                xm, hm = self.mol_conv(xm, data_mol.edge_index, data_mol.edge_attr, h=hm, batch=data_mol.batch)
            xp, hp = self.pro_conv(xp, data_pro.edge_index, data_pro.edge_attr, h=hp, batch=data_pro.batch)
            # (every step's outputs feed this fusion AND the next step / the readouts: they come back from the fusion node, so that both
            #  gradients meet inside its backward launch instead of in an add launch per tower)
            f, xm, xp = dot_and_global_pool2(xm, xp, data_mol.batch, data_pro.batch, with_identity=True)
            fusion.append(f)
        nm = getattr(data_mol, "num_graphs", None) or None
        np_ = getattr(data_pro, "num_graphs", None) or None
        outm = self.mol_flat(self.mol_readout(xm, data_mol.batch, nm))
        outp = self.pro_flat(self.pro_readout(xp, data_pro.batch, np_))
        out = ops.cat_cols([outm, outp] + fusion)      # (model.py:74-75; contiguous gradients for every piece from one launch)
        return self.lin_out1(self.lin_out0(out))


class ArchitectureDDI(torch.nn.Module):
    """Two-drug model: two ligand towers with their own parameters and per-pair fusion (src_2gi_ddi/model.py:9-62; checkpoint keys
    ``mol1_lin0.*``, ``mol2_lin0.*``, ``mol1_conv.*``, ``mol2_conv.*``, ``mol1_readout.*``, ``mol2_readout.*``, ``mol1_flat.*``,
    ``mol2_flat.*``, ``lin_out0.*``, ``lin_out1.*``; parameter creation order = the reference's, so seeded inits agree)."""

    def __init__(self, mol_in_dim=15, mol_edge_in_dim=4, hid_dim_alpha=4, e_dim=1024, out_dim=1, mol_block="_NNConv", message_steps=3,
                 mol_readout="GlobalPool5",
                 pre_norm="_None", graph_norm="_None", flat_norm="_None", end_norm="_None",
                 pre_do="_None()", graph_do="Dropout(0.2)", flat_do="_None()", end_do="Dropout(0.2)",
                 pre_act="RReLU", graph_act="RReLU", flat_act="RReLU", end_act="RReLU", graph_res=True):
        super().__init__()
        hid_dim = mol_in_dim * hid_dim_alpha
        self.mol1_lin0 = LinearBlock(mol_in_dim, hid_dim, norm=pre_norm, dropout=pre_do, act=pre_act)
        self.mol2_lin0 = LinearBlock(mol_in_dim, hid_dim, norm=pre_norm, dropout=pre_do, act=pre_act)
        self.mol1_conv = MessageBlock(hid_dim, hid_dim, mol_edge_in_dim, norm=graph_norm, dropout=graph_do, conv=mol_block,
                                      act=graph_act, res=graph_res)
        self.mol2_conv = MessageBlock(hid_dim, hid_dim, mol_edge_in_dim, norm=graph_norm, dropout=graph_do, conv=mol_block,
                                      act=graph_act, res=graph_res)
        self.message_steps = message_steps
        self.mol1_readout = _readout(mol_readout, hid_dim)
        self.mol2_readout = _readout(mol_readout, hid_dim)
        _mol_ro = 5 if mol_readout == "GlobalPool5" else 2
        self.mol1_flat = LinearBlock(_mol_ro * hid_dim, hid_dim, norm=flat_norm, dropout=flat_do, act=flat_act)
        self.mol2_flat = LinearBlock(_mol_ro * hid_dim, hid_dim, norm=flat_norm, dropout=flat_do, act=flat_act)
        self.lin_out0 = LinearBlock(hid_dim * 2 + message_steps * 2, e_dim, norm=end_norm, dropout=end_do, act=end_act)
        self.lin_out1 = LinearBlock(e_dim, out_dim, norm=end_norm, dropout=end_do, act="_None")

    def forward(self, mol1, mol2):
        return graphs.graphed_call(self, self._eager_forward, mol1, mol2)              # (see Architecture.forward)

    def _eager_forward(self, mol1, mol2):
        with ops.weight_scope():
            return self._forward(mol1, mol2)

    def _forward(self, mol1, mol2):
        prestage_pass((self.mol1_lin0, self.mol1_conv, mol1.x, mol1.edge_attr), (self.mol2_lin0, self.mol2_conv, mol2.x, mol2.edge_attr))
        x1 = self.mol1_lin0(mol1.x, batch=mol1.batch)
        x2 = self.mol2_lin0(mol2.x, batch=mol2.batch)
        h1, h2 = None, None
        fusion = []
        for i in range(self.message_steps):
            with ops.block_feeds_itself(i + 1 < self.message_steps):      # (the fusion hands the rows back untouched)
                x1, h1 = self.mol1_conv(x1, mol1.edge_index, mol1.edge_attr, h=h1, batch=mol1.batch)
                x2, h2 = self.mol2_conv(x2, mol2.edge_index, mol2.edge_attr, h=h2, batch=mol2.batch)
            f, x1, x2 = dot_and_global_pool2(x1, x2, mol1.batch, mol2.batch, with_identity=True)
            fusion.append(f)
        n1 = getattr(mol1, "num_graphs", None) or None
        n2 = getattr(mol2, "num_graphs", None) or None
        o1 = self.mol1_flat(self.mol1_readout(x1, mol1.batch, n1))
        o2 = self.mol2_flat(self.mol2_readout(x2, mol2.batch, n2))
        out = ops.cat_cols([o1, o2] + fusion)
        return self.lin_out1(self.lin_out0(out))
