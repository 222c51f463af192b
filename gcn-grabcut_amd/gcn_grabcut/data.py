"""
Minimal graph containers standing in for torch_geometric.data.{Data, Batch}.

The reference hands its models "any object with .x, .edge_index, .edge_attr and
an optional .batch" (reference model.py:509-513); PyG itself is not a dependency
of this build.  `Batch.from_data_list` follows PyG's collation:
concatenate x / edge_attr, offset edge_index by the cumulative node counts, and
record the graph id of every node.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch


class Data:
    def __init__(self, x: Optional[torch.Tensor] = None, edge_index: Optional[torch.Tensor] = None,
                 edge_attr: Optional[torch.Tensor] = None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        self.edge_attr = edge_attr
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        return 0 if self.x is None else int(self.x.size(0))

    @property
    def num_edges(self) -> int:
        return 0 if self.edge_index is None else int(self.edge_index.size(1))

    def _apply(self, fn):
        out = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            out.__dict__[k] = fn(v) if torch.is_tensor(v) else v
        return out

    def to(self, device, non_blocking: bool = False):
        return self._apply(lambda t: t.to(device, non_blocking=non_blocking))

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def cpu(self):
        return self.to("cpu")

    def __repr__(self) -> str:
        parts = [f"{k}={list(v.shape) if torch.is_tensor(v) else v}" for k, v in self.__dict__.items()]
        return f"{self.__class__.__name__}({', '.join(parts)})"


class Batch(Data):
    """Several graphs as one disconnected graph; `batch[i]` is node i's graph id."""

    @classmethod
    def from_data_list(cls, graphs: Sequence[Data]) -> "Batch":
        if not graphs:
            raise ValueError("from_data_list needs at least one graph")
        xs, eis, eas, bs = [], [], [], []
        ptr = [0]
        for gid, g in enumerate(graphs):
            n = g.x.size(0)
            xs.append(g.x)
            eis.append(g.edge_index + ptr[-1])
            if g.edge_attr is not None:
                eas.append(g.edge_attr)
            bs.append(torch.full((n,), gid, dtype=torch.long, device=g.x.device))
            ptr.append(ptr[-1] + n)
        out = cls(
            x=torch.cat(xs, 0),
            edge_index=torch.cat(eis, 1),
            edge_attr=torch.cat(eas, 0) if eas else None,
        )
        out.batch = torch.cat(bs, 0)
        out.ptr = torch.tensor(ptr, dtype=torch.long, device=out.x.device)
        out.num_graphs = len(graphs)
        return out
