"""
GCN-GrabCut on MI355X — host mirror of the reference package src/gcn_grabcut.

Same public names as the reference (__init__.py:57-81) for the per-image
segmentation hot path; all arithmetic runs in libggc_hip.so (hand-written
gfx950 kernels behind the C ABI of include/ggc.h).  `dataset` holds the
graph-cache writer behind tools/prepare_graphs.py;
training, loaders and plotting are out of scope.
"""
from ._constants import N_NODE_FEATS, N_EDGE_FEATS, N_PRIOR_FEATS, N_IMAGE_FEATS
from .data import Data, Batch
from .grabcut import GrabCut, GrabCutConfig, Label
from .graph_builder import (
    GraphBuilder, SuperpixelGraph, SuperpixelGraphConfig, compute_auto_prior, encode_user_hints,
)
from .metrics import evaluate, evaluate_batch, evaluate_trimap, boundary_f1, SegmentationMetrics, TrimapMetrics
from .model import (
    ResGCNNet, GCNTrimapNet, GATTrimapNet, build_model, _probs_to_trimap, probs_to_node_trimap, project_to_pixels,
    TRIMAP_BG, TRIMAP_FG, TRIMAP_PROB_BG, TRIMAP_PROB_FG, CLASS_BG, CLASS_UNK, CLASS_FG,
)
from .pipeline import GCNGrabCutPipeline, SegmentationResult, clean_mask, guided_filter, refine_trimap
from .synthetic import synthetic_image, synthetic_batch

__version__ = "0.3.0+mi355x.1"

__all__ = [
    "GrabCut", "GrabCutConfig", "Label",
    "GraphBuilder", "SuperpixelGraph", "SuperpixelGraphConfig", "compute_auto_prior", "encode_user_hints",
    "N_NODE_FEATS", "N_EDGE_FEATS", "N_PRIOR_FEATS",
    "evaluate", "evaluate_batch", "evaluate_trimap", "boundary_f1", "SegmentationMetrics", "TrimapMetrics",
    "GCNGrabCutPipeline", "SegmentationResult", "clean_mask", "guided_filter", "refine_trimap",
    "ResGCNNet", "GCNTrimapNet", "GATTrimapNet", "build_model", "probs_to_node_trimap", "project_to_pixels",
    "Data", "Batch", "synthetic_image", "synthetic_batch",
]
