"""
Superpixel graph construction — host mirror of reference src/gcn_grabcut/graph_builder.py.

Same public names (SuperpixelGraphConfig, SuperpixelGraph, GraphBuilder,
compute_auto_prior, encode_user_hints, N_* constants); the arithmetic — colour
conversion, SLIC, region statistics, adjacency and non-local edges, automatic
prior — runs on the MI355X through libggc_hip.so (ggc_preprocess, ggc_slic,
ggc_graph_count / ggc_graph_fill).

Node features (16): mean Lab, std Lab (per-image min-max), mean HSV, centroid y/x,
area ratio, compactness, mean gradient, boundary ratio, centre distance.
Edge features (5): colour distance, centroid distance, shared boundary, gradient
contrast, non-local flag.  Prior (3): fg-ness, bg-ness, ambiguity.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from ._constants import N_IMAGE_FEATS, N_PRIOR_FEATS, N_HINT_FEATS, N_NODE_FEATS, N_EDGE_FEATS  # noqa: F401


@dataclass
class SuperpixelGraphConfig:
    """reference graph_builder.py:64-71"""
    n_segments: int = 300       # target superpixel count
    compactness: float = 10.0   # SLIC spatial regularisation
    sigma: float = 1.0          # Gaussian pre-smoothing
    use_lab: bool = True        # SLIC on the Lab image
    connectivity: int = 4       # 4 or 8 — pixel adjacency for edge detection
    n_nonlocal: int = 4         # non-local colour neighbours per node (0 = off)


@dataclass
class SuperpixelGraph:
    """Container for a built superpixel graph (reference graph_builder.py:80-129)."""
    segments: np.ndarray        # (H, W) int32
    node_features: np.ndarray   # (N, 16) float32
    edge_index: np.ndarray      # (2, E) int64, symmetric directed pairs
    edge_attr: np.ndarray       # (E, 5) float32
    n_nodes: int = 0
    n_edges: int = 0
    node_centroids: np.ndarray = field(default_factory=lambda: np.empty((0, 2)))
    prior_features: np.ndarray = field(default_factory=lambda: np.empty((0, N_PRIOR_FEATS)))
    node_areas: np.ndarray = field(default_factory=lambda: np.empty((0,)))

    def node_input(self, prior_features: np.ndarray | None = None) -> np.ndarray:
        """(N, 19) = image features || automatic prior."""
        prior = self.prior_features if prior_features is None else prior_features
        if prior is None or prior.size == 0:
            prior = np.zeros((self.n_nodes, N_PRIOR_FEATS), dtype=np.float32)
        return np.concatenate([self.node_features, prior], axis=1).astype(np.float32)

    def to_networkx(self):
        import networkx as nx
        g = nx.Graph()
        g.add_nodes_from(range(self.n_nodes))
        nx.set_node_attributes(g, {i: self.node_features[i] for i in range(self.n_nodes)}, "feat")
        for i in range(self.edge_index.shape[1]):
            s, d = self.edge_index[0, i], self.edge_index[1, i]
            if s < d:
                g.add_edge(int(s), int(d), attr=self.edge_attr[i])
        return g

    def to_pyg(self, prior_features: np.ndarray | None = None):
        """Graph as this package's Data container (PyG itself is not a dependency)."""
        import torch
        from .data import Data
        area = self.node_areas
        if area is None or area.size == 0:
            area = np.full(self.n_nodes, 1.0 / max(self.n_nodes, 1), dtype=np.float32)
        return Data(
            x=torch.tensor(self.node_input(prior_features), dtype=torch.float32),
            edge_index=torch.tensor(self.edge_index, dtype=torch.long),
            edge_attr=torch.tensor(self.edge_attr, dtype=torch.float32),
            node_area=torch.tensor(area, dtype=torch.float32),
        )


def _check_image(image: np.ndarray) -> np.ndarray:
    if not isinstance(image, np.ndarray) or image.ndim != 3 or image.shape[2] != 3 or image.dtype != np.uint8:
        raise ValueError("image must be a BGR uint8 array of shape (H, W, 3)")
    return np.ascontiguousarray(image)


def graphs_to_host(graphs, index: int = 0) -> SuperpixelGraph:
    """One image of a DeviceGraphs batch as the reference's host container."""
    n0, n1 = int(graphs.node_ptr_host[index]), int(graphs.node_ptr_host[index + 1])
    e0, e1 = int(graphs.edge_ptr_host[index]), int(graphs.edge_ptr_host[index + 1])
    x = graphs.x[n0:n1].cpu().numpy()
    src = graphs.edge_src[e0:e1].cpu().numpy().astype(np.int64) - n0
    dst = graphs.edge_dst[e0:e1].cpu().numpy().astype(np.int64) - n0
    return SuperpixelGraph(
        segments=graphs.segments[index].cpu().numpy(),
        node_features=np.ascontiguousarray(x[:, :N_IMAGE_FEATS]),
        edge_index=np.stack([src, dst]) if e1 > e0 else np.zeros((2, 0), np.int64),
        edge_attr=graphs.edge_attr[e0:e1].cpu().numpy().reshape(-1, N_EDGE_FEATS),
        n_nodes=n1 - n0,
        n_edges=e1 - e0,
        node_centroids=graphs.centroids[n0:n1].cpu().numpy(),
        prior_features=np.ascontiguousarray(x[:, N_IMAGE_FEATS:]),
        node_areas=graphs.area_ratio[n0:n1].cpu().numpy(),
    )


class GraphBuilder:
    """
    Builds the superpixel adjacency graph of a BGR image on the MI355X
    (reference graph_builder.py:131-175).

        graph = GraphBuilder(image, SuperpixelGraphConfig(n_segments=500)).build()
    """

    def __init__(self, image: np.ndarray, config: SuperpixelGraphConfig | None = None, device="cuda"):
        from ._engine import get_engine
        self.bgr = _check_image(image)
        self.config = config or SuperpixelGraphConfig()
        if self.config.connectivity not in (4, 8):
            raise ValueError("connectivity must be 4 or 8")
        self._eng = get_engine(device)
        self._bgr_d = self._eng.to_device(self.bgr[None])
        # colour prep happens here, like the reference constructor (:142-154)
        self._lab, self._hsv, self._gray, self._grad = self._eng.preprocess(self._bgr_d)

    def build_device(self):
        cfg = self.config
        if cfg.use_lab:
            seg, n = self._eng.slic(self._lab, cfg.n_segments, cfg.compactness, cfg.sigma)
        else:                                      # reference graph_builder.py:177-179: slic(self.rgb.astype(float), ...)
            seg, n = self._eng.slic_rgb(self._bgr_d, cfg.n_segments, cfg.compactness, cfg.sigma)
        return self._eng.build_graphs(seg, n, self._lab, self._hsv, self._grad, cfg.connectivity, cfg.n_nonlocal)

    def build(self) -> SuperpixelGraph:
        return graphs_to_host(self.build_device(), 0)


def compute_auto_prior(segments: np.ndarray, lab: np.ndarray, centre_sigma: float = 0.45,
                       contrast_sigma: float = 0.40, device="cuda") -> np.ndarray:
    """(N, 3) float32 [fg-ness, bg-ness, ambiguity] — reference graph_builder.py:357-444."""
    import torch
    from ._engine import Engine, get_engine
    eng = get_engine(device)
    custom = (float(centre_sigma), float(contrast_sigma)) != (0.45, 0.40)
    if custom:
        # The sigmas are state of a library context.  Non-default values go to a PRIVATE context (one per device, serialised
        # by a lock), so the shared context — which other threads build the pipeline's graphs with — never leaves the defaults.
        with _prior_lock:
            peng = _prior_engines.get(eng.index)
            if peng is None:
                peng = _prior_engines[eng.index] = Engine(eng.index, private_context=True)
            peng.ctx.call("ggc_graph_prior_sigmas", float(centre_sigma), float(contrast_sigma))
            return _auto_prior_on(peng, segments, lab)
    return _auto_prior_on(eng, segments, lab)


_prior_engines: dict = {}
_prior_lock = __import__("threading").Lock()


def _auto_prior_on(eng, segments: np.ndarray, lab: np.ndarray) -> np.ndarray:
    import torch
    seg = eng.to_device(np.ascontiguousarray(segments, dtype=np.int32)[None])
    lab_d = eng.to_device(np.ascontiguousarray(lab, dtype=np.float32)[None])
    h, w = segments.shape
    n = torch.tensor([int(segments.max()) + 1], dtype=torch.int32, device=eng.device)
    zeros3 = torch.zeros(1, h, w, 3, device=eng.device)
    zeros1 = torch.zeros(1, h, w, device=eng.device)
    graphs = eng.build_graphs(seg, n, lab_d, zeros3, zeros1, 4, 0)
    return graphs.x[:, N_IMAGE_FEATS:].cpu().numpy()


def encode_user_hints(segments: np.ndarray, fg_points, bg_points) -> np.ndarray:
    """Legacy click features (reference graph_builder.py:457-494); a host-side
    table lookup that the automatic pipeline never calls."""
    n_nodes = int(segments.max()) + 1
    hints = np.zeros((n_nodes, 3), dtype=np.float32)
    hints[:, 2] = 1.0
    for col, points in ((0, fg_points), (1, bg_points)):
        for r, c in points:
            r, c = int(r), int(c)
            if 0 <= r < segments.shape[0] and 0 <= c < segments.shape[1]:
                nid = int(segments[r, c])
                hints[nid, col] = 1.0
                hints[nid, 2] = 0.0
    return hints
