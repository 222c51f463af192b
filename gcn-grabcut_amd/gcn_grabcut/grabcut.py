"""
GrabCut — host mirror of reference src/gcn_grabcut/grabcut.py.

Same class and method names; cv2.grabCut is replaced by ggc_grabcut in
libggc_hip.so (GMM colour models + push-relabel min-cut on the 8-neighbour
pixel grid, on the MI355X).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import IntEnum
from typing import List, Optional, Tuple

import numpy as np


class Label(IntEnum):
    """Pixel label constants — OpenCV GrabCut convention (reference grabcut.py:22-27)."""
    BG_DEFINITE = 0   # cv2.GC_BGD
    FG_DEFINITE = 1   # cv2.GC_FGD
    BG_PROBABLE = 2   # cv2.GC_PR_BGD
    FG_PROBABLE = 3   # cv2.GC_PR_FGD


@dataclass
class GrabCutConfig:
    """reference grabcut.py:30-35.  As in the reference, only n_iter and color_space
    take effect (gamma and n_components are fixed at OpenCV's 50 and 5)."""
    n_iter: int = 5
    n_components: int = 5
    gamma: float = 50.0
    color_space: str = "rgb"   # "rgb" | "hsv" | "lab"
    seed: int = 0              # additive: seed of the k-means++ GMM initialisation


@dataclass
class GrabCutSnapshot:
    """State snapshot captured after each GrabCut run."""
    tag: str
    fg_pixels: int
    bg_pixels: int
    fg_ratio: float
    mask_copy: np.ndarray = field(repr=False)


class GrabCut:
    """
    gc = GrabCut(image)
    mask = gc.run_with_bbox((x, y, w, h))    # classical mode
    mask = gc.run_with_trimap(trimap)        # GCN-guided mode
    """

    def __init__(self, image: np.ndarray, config: Optional[GrabCutConfig] = None, device="cuda"):
        from ._engine import get_engine
        if not isinstance(image, np.ndarray) or image.ndim != 3 or image.shape[2] != 3 or image.dtype != np.uint8:
            raise ValueError("image must be a BGR uint8 array of shape (H, W, 3)")
        self.image = image
        self.config = config or GrabCutConfig()
        self.mask: Optional[np.ndarray] = None
        self._bgd = np.zeros((1, 65), np.float64)
        self._fgd = np.zeros((1, 65), np.float64)
        self.history: List[GrabCutSnapshot] = []
        self._eng = get_engine(device)
        self._proc = self._preprocess(image)

    def _preprocess(self, image: np.ndarray):
        """BGR uint8 -> the (1,H,W,3) uint8 device image GrabCut works on (reference grabcut.py:73-79).  hsv / lab are
        8-bit conversions in the style of cv2.cvtColor (ggc_convert_color8; parity with OpenCV unpinned)."""
        cs = self.config.color_space.lower()
        dev = self._eng.to_device(np.ascontiguousarray(image)[None])
        if cs in ("hsv", "lab"):
            return self._eng.convert_color8(dev, cs)
        if cs != "rgb":
            raise ValueError(f"unknown color_space '{cs}': rgb | hsv | lab")
        return dev

    def _run(self, mask: Optional[np.ndarray], n_iter: int, mode: int, rect=None) -> np.ndarray:
        import torch
        eng = self._eng
        h, w = self.image.shape[:2]
        dmask = eng.to_device(mask[None]) if mask is not None else torch.zeros(1, h, w, dtype=torch.uint8, device=eng.device)
        bgd, fgd = eng.to_device(self._bgd), eng.to_device(self._fgd)
        _, dmask, bgd, fgd = eng.grabcut(self._proc, dmask, n_iter, mode, None if rect is None else [list(rect)],
                                         self.config.seed, bgd, fgd)
        self.mask = dmask[0].cpu().numpy()
        self._bgd, self._fgd = bgd.cpu().numpy(), fgd.cpu().numpy()
        return self._binary()

    def run_with_bbox(self, bbox: Tuple[int, int, int, int]) -> np.ndarray:
        """Classical GrabCut, bbox = (x, y, w, h) — reference grabcut.py:81-102."""
        self._bgd = np.zeros((1, 65), np.float64)
        self._fgd = np.zeros((1, 65), np.float64)
        out = self._run(None, self.config.n_iter, 1, bbox)
        self._snapshot("bbox_init")
        return out

    def run_with_trimap(self, trimap: np.ndarray) -> np.ndarray:
        """GCN-guided GrabCut seeded with a trimap in {0,1,2,3} — reference grabcut.py:104-151.
        The promotion of probable labels and the single-class guard run inside ggc_grabcut."""
        if trimap.shape != self.image.shape[:2]:
            raise ValueError(f"Trimap shape {trimap.shape} != image shape {self.image.shape[:2]}")
        if trimap.dtype != np.uint8:
            trimap = trimap.astype(np.uint8)
        self._bgd = np.zeros((1, 65), np.float64)
        self._fgd = np.zeros((1, 65), np.float64)
        out = self._run(np.ascontiguousarray(trimap), self.config.n_iter, 0)
        degenerate = not ((self.mask == Label.FG_DEFINITE).any() and (self.mask == Label.BG_DEFINITE).any())
        self._snapshot("trimap_degenerate" if degenerate else "trimap_init")
        return out

    def refine(self, extra_iter: int = 3) -> np.ndarray:
        """Continue from the current GMM state (GC_EVAL) — reference grabcut.py:153-163."""
        if self.mask is None:
            raise RuntimeError("Call run_with_bbox or run_with_trimap first.")
        out = self._run(self.mask, extra_iter, 2)
        self._snapshot("refinement")
        return out

    def _binary(self) -> np.ndarray:
        return np.where((self.mask == Label.FG_DEFINITE) | (self.mask == Label.FG_PROBABLE), 1, 0).astype(np.uint8)

    def _snapshot(self, tag: str) -> None:
        b = self._binary()
        self.history.append(GrabCutSnapshot(tag=tag, fg_pixels=int(b.sum()), bg_pixels=int((b == 0).sum()),
                                            fg_ratio=float(b.mean()), mask_copy=self.mask.copy()))

    def _compose(self, alpha: float, color: Tuple):
        eng = self._eng
        binary = eng.to_device(self._binary()[None])
        return eng.compose(eng.to_device(np.ascontiguousarray(self.image)[None]), binary, alpha, tuple(color[::-1]))

    def overlay_mask(self, alpha: float = 0.45, color: Tuple = (0, 220, 100)) -> np.ndarray:
        """BGR image with a coloured foreground overlay — reference grabcut.py:180-188."""
        return self._compose(alpha, color)[0][0].cpu().numpy()

    def crop_foreground(self) -> np.ndarray:
        """BGRA image with the background transparent — reference grabcut.py:190-195."""
        return self._compose(0.45, (0, 220, 100))[1][0].cpu().numpy()

    def trimap_visualisation(self, trimap: np.ndarray) -> np.ndarray:
        vis = np.zeros((*trimap.shape, 3), dtype=np.uint8)
        vis[trimap == Label.BG_DEFINITE] = [0, 0, 0]
        vis[trimap == Label.FG_DEFINITE] = [255, 255, 255]
        vis[trimap == Label.BG_PROBABLE] = [80, 0, 0]
        vis[trimap == Label.FG_PROBABLE] = [0, 200, 200]
        return vis
