"""argparse groups with the reference's flag names and defaults (subgraph_counting/config.py:185-400;
defaults pinned by tests/golden/config_defaults.json).  ``main.py`` splits the parsed values into
three Namespaces by stripping the ``neigh_`` / ``gossip_`` prefixes (main.py:539-558);
``split_namespaces`` does the same."""
from __future__ import annotations

import argparse

_NEIGH = [  # (flag, type or "flag", default, help)
    ("--neigh_conv_type", str, "SAGE", "type of convolution"),
    ("--neigh_layer_num", int, 8, "Number of graph conv layers"),
    ("--neigh_input_dim", int, 1, "Training input size"),
    ("--neigh_hidden_dim", int, 64, "Training hidden size"),
    ("--neigh_dropout", float, 0.0, "Dropout rate"),
    ("--neigh_model_path", str, "ckpt/DeSCo/Syn_1827/neigh", "path to save/load model"),
    ("--neigh_epoch_num", int, 300, "number of training epochs"),
    ("--neigh_batch_size", int, 512, "batch size"),
    ("--depth", int, 4, "depth of the canonical neighborhood"),
    ("--use_hetero", "flag", True, "whether to use heterogeneous GNNs"),
    ("--use_tconv", "flag", True, "whether to use triangle convolution (SHMP)"),
    ("--use_canonical", "flag", True, "whether to use canonical partition"),
    ("--use_node_feature", "flag", False, "whether to use node features"),
    ("--neigh_weight_decay", float, 0.0, "weight decay"),
    ("--neigh_lr", float, 1e-4, "learning rate"),
    ("--neigh_tune_lr", "flag", False, "auto tune learning rate"),
    ("--neigh_tune_bs", "flag", False, "auto tune batch size"),
    ("--zero_node_feat", "flag", False, "zero the node features"),
]
_GOSSIP = [
    ("--gossip_conv_type", str, "GOSSIP", "type of convolution"),
    ("--gossip_layer_num", int, 2, "Number of graph conv layers"),
    ("--gossip_hidden_dim", int, 64, "Training hidden size"),
    ("--gossip_dropout", float, 0.01, "Dropout rate"),
    ("--gossip_model_path", str, "ckpt/DeSCo/Syn_1827/gossip", "path to save/load model"),
    ("--gossip_epoch_num", int, 30, "number of training epochs"),
    ("--gossip_batch_size", int, 256, "batch size"),
    ("--gossip_lr", float, 1e-3, "learning rate"),
    ("--weight_decay", float, 0.0, "weight decay"),
    ("--gossip_tune_lr", "flag", False, "auto tune learning rate"),
    ("--gossip_tune_bs", "flag", False, "auto tune batch size"),
]
_OPT = [
    ("--train_dataset", str, "Syn_1827", "name of the training dataset"),
    ("--valid_dataset", str, "Syn_1827", "name of the validation dataset"),
    ("--test_dataset", str, "MUTAG", "name of the test dataset"),
    ("--gpu", "ints", 0, "the id of gpus to use, support multi-gpu"),
    ("--num_cpu", int, 8, "number of cpu to use"),
    ("--output_dir", str, None, "path to save raw output"),
    ("--neigh_checkpoint", str, None, "path to load neighborhood counting model"),
    ("--gossip_checkpoint", str, None, "path to load gossip counting model"),
    ("--train_neigh", "flag", False, "train the neighborhood counting model"),
    ("--train_gossip", "flag", False, "train the gossip counting model"),
    ("--test_gossip", "flag", False, "run gossip counting at test time"),
]


def _add(parser, title, table):
    grp = parser.add_argument_group(title)
    defaults = {}
    for flag, typ, default, help_ in table:
        if typ == "flag":
            grp.add_argument(flag, action="store_true", help=help_)
        elif typ == "ints":
            grp.add_argument(flag, nargs="+", type=int, help=help_)
        else:
            grp.add_argument(flag, type=typ, help=help_)
        defaults[flag.lstrip("-")] = default
    grp.set_defaults(**defaults)
    return grp._group_actions


def parse_neighborhood(parser, arg_str=None):
    return _add(parser, "neighborhood counting model arguments", _NEIGH)


def parse_gossip(parser, arg_str=None):
    return _add(parser, "gossip counting model arguments", _GOSSIP)


def parse_optimizer(parser):
    return _add(parser, "optimizer arguments", _OPT)


def split_namespaces(args: argparse.Namespace):
    """(args_neighborhood, args_gossip, args_opt) with prefixes stripped, as main.py:539-558."""
    def pick(table, prefix):
        ns = argparse.Namespace()
        for flag, *_ in table:
            k = flag.lstrip("-")
            setattr(ns, k[len(prefix):] if k.startswith(prefix) else k, getattr(args, k))
        return ns
    return pick(_NEIGH, "neigh_"), pick(_GOSSIP, "gossip_"), pick(_OPT, "")
