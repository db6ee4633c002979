"""Equalisation work list: which (layer, next layer) pairs can trade per-channel scales.

Host-side graph logic with the interface of the reference's ``utils/relation.py`` (``Relation`` :5-27,
``create_relation`` :30-94); it only builds the list the HIP engine consumes.

A pair (A, B) is formed when B's single input chain leads straight up to A through nothing but
scale-commuting nodes (BatchNorm2d, ReLU, QuantMeasure, AvgPool2d, ``F.pad``, ``torch.mean``) and
every node on the way -- A included -- feeds exactly one consumer.  A layer claimed as predecessor
by two successors is dropped from the list (relation.py:62-67).
"""
from __future__ import annotations

from collections import OrderedDict

import torch.nn as nn


class Relation:
    """(layer_first, layer_second, bn between them) plus the cumulative per-channel scale S."""

    def __init__(self, layer_idx_1, layer_idx_2, bn_idx_1):
        self.layer_first = layer_idx_1
        self.layer_second = layer_idx_2
        self.bn_idx = bn_idx_1
        self.S = None

    def __repr__(self):
        return '({}, {})'.format(self.layer_first, self.layer_second)

    def get_idxs(self):
        return self.layer_first, self.layer_second, self.bn_idx

    def set_scale_vec(self, S):
        # cumulative product over sweeps (relation.py:20-24)
        if self.S is None:
            self.S = S
        else:
            self.S *= S

    def get_scale_vec(self):
        return self.S


def _passthrough_types():
    from .quantize import QuantMeasure
    return (nn.BatchNorm2d, nn.ReLU, QuantMeasure, nn.AvgPool2d)


def _fanout(graph, bottoms):
    out = {}
    for key in graph:
        if key == 'Data':
            continue
        for b in bottoms[key]:
            out[b] = out.get(b, 0) + 1
    return out


def _walk_up(graph, bottoms, key, targ_type, fanout, passthrough):
    """Return (previous targ layer key, last BN key seen on the way) or (None, None)."""
    bn_seen = None
    bot = bottoms[key]
    while len(bot) == 1 and bot[0] != 'Data' and fanout[bot[0]] == 1:
        up = bot[0]
        node = graph[up]
        if type(node) == nn.BatchNorm2d:
            bn_seen = up
        if type(node) in targ_type:
            return up, bn_seen
        commuting = type(node) in passthrough or (
            isinstance(node, str) and ('F.pad' in up or 'torch.mean' in up))
        if not commuting:
            break
        bot = bottoms[up]
    return None, None


def create_relation(graph, bottoms, targ_type=None, delete_single=False):
    if targ_type is None:
        from .quantize import QConv2d
        targ_type = [QConv2d]
    passthrough = _passthrough_types()
    fanout = _fanout(graph, bottoms)

    found = OrderedDict()
    for key in graph:
        if type(graph[key]) not in targ_type:
            continue
        prev, bn = _walk_up(graph, bottoms, key, targ_type, fanout, passthrough)
        if prev in found:
            del found[prev]                    # claimed twice: not a 1-to-1 chain
        elif prev is not None:
            found[prev] = Relation(prev, key, bn)
    rels = list(found.values())
    if not delete_single:
        return rels

    # keep only chains of >= 2 relations (conv->conv->conv); relation.py:69-90
    chains = []
    for rr in rels:
        home = None
        for chain in chains:
            if any(rr.layer_first == other.layer_second for other in chain):
                home = chain
        if home is None:
            chains.append([rr])
        else:
            home.append(rr)
    return [rr for chain in chains if len(chain) > 1 for rr in chain]
