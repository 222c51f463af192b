#!/opt/conda/bin/python3.9
"""
Generates tests/golden/skimage_*.npz with the third-party oracles that exist in
the build container only (scikit-image 0.18.3 + scipy 1.7.1 under
/opt/conda/bin/python3.9).  Run from the repo root:

    /opt/conda/bin/python3.9 tests/golden/make_golden_skimage.py

The fixtures are data (inputs and library outputs); nothing of the reference
repository is involved.  They pin the C oracle in oracle/{color,slic,graph}.c.
"""
import importlib.util
import os
import sys

import numpy as np
import scipy
import skimage
from scipy import ndimage as ndi
from skimage.color import rgb2hsv, rgb2lab
from skimage.segmentation import find_boundaries, slic
from skimage.segmentation._slic import _enforce_label_connectivity_cython, _slic_cython
from skimage.util import regular_grid

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
spec = importlib.util.spec_from_file_location(
    "synthetic", os.path.join(ROOT, "gcn-grabcut_amd", "gcn_grabcut", "synthetic.py"))
synthetic = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synthetic)

CASES = [  # (H, W, n_segments, seed)
    (64, 64, 50, 1), (72, 96, 120, 2), (96, 128, 200, 3), (50, 81, 300, 4), (96, 96, 40, 5),
]


def slic_case(h, w, n_segments, seed, compactness=10.0, sigma=1.0):
    bgr = synthetic.synthetic_image(h, w, seed)
    rgb = np.ascontiguousarray(bgr[:, :, ::-1])
    lab64 = rgb2lab(rgb)
    lab = lab64.astype(np.float32)
    hsv = rgb2hsv(rgb).astype(np.float32)

    # --- what skimage >= 0.19 does before the 0.18.3 body: global min-max rescale (emulated)
    x = lab.copy()
    x -= x.min()
    x /= x.max()
    # --- 0.18.3 body, step by step, on the rescaled input
    img = x[np.newaxis, ...]                        # (1,H,W,3) float32
    img_lab = rgb2lab(img)                          # second Lab, float32
    assert img_lab.dtype == np.float32
    slices = regular_grid(img_lab.shape[:3], n_segments)
    gz, gy, gx = np.mgrid[:1, :h, :w]
    cent = np.concatenate([g[slices].ravel()[..., None] for g in (gz, gy, gx)], axis=-1)
    steps = np.asarray([float(s.step) if s.step is not None else 1.0 for s in slices])
    sig = np.array([sigma, sigma, sigma], dtype=np.float32)
    smoothed = ndi.gaussian_filter(img_lab, list(sig) + [0])
    assert smoothed.dtype == np.float32
    k = cent.shape[0]
    segments0 = np.ascontiguousarray(np.concatenate([cent, np.zeros((k, 3))], axis=-1), dtype=np.float32)
    step = float(max(steps))
    scaled = np.ascontiguousarray(smoothed * (1.0 / compactness), dtype=np.float32)
    seg_work = segments0.copy()
    raw = _slic_cython(scaled, None, seg_work, step, 10, np.ones(3, np.float32), False,
                       ignore_color=False, start_label=0)
    seg_size = np.prod(scaled.shape[:3]) / k
    min_size, max_size = int(0.5 * seg_size), int(3 * seg_size)
    conn = _enforce_label_connectivity_cython(np.ascontiguousarray(raw), min_size, max_size, start_label=0)
    whole = slic(x, n_segments=n_segments, compactness=compactness, sigma=sigma, start_label=0,
                 multichannel=True)
    assert np.array_equal(whole, conn[0])
    bnd = find_boundaries(conn[0], mode="inner")
    return dict(
        bgr=bgr, lab=lab, hsv=hsv, n_segments=np.int64(n_segments),
        rescaled=x, second_lab=img_lab[0], smoothed=smoothed[0], scaled=scaled[0],
        seeds=segments0[:, 1:3].copy(), step=np.float32(step), centers_final=seg_work,
        raw=raw[0].astype(np.int32), connected=conn[0].astype(np.int32),
        min_size=np.int64(min_size), max_size=np.int64(max_size), boundaries=bnd.astype(np.uint8),
    )


def main():
    out = {}
    for i, (h, w, n, seed) in enumerate(CASES):
        for key, val in slic_case(h, w, n, seed).items():
            out[f"c{i}_{key}"] = val
    # a connectivity stress case: random blobs with many tiny and a few huge components
    rs = np.random.RandomState(7)
    lab = (rs.rand(48, 64) * 6).astype(np.intp)
    lab[10:40, 5:60] = 9
    lab[20:22, 20:50] = 3
    for mn, mx, tag in ((4, 200, "a"), (12, 60, "b"), (1, 10 ** 6, "c")):
        r = _enforce_label_connectivity_cython(np.ascontiguousarray(lab[None]), mn, mx, start_label=0)
        out[f"stress_{tag}"] = r[0].astype(np.int32)
    out["stress_in"] = lab.astype(np.int32)
    out["versions"] = np.array([skimage.__version__, scipy.__version__, np.__version__, sys.version.split()[0]])
    np.savez_compressed(os.path.join(HERE, "skimage_0183.npz"), **out)
    print("wrote", os.path.join(HERE, "skimage_0183.npz"), "cases:", len(CASES))


if __name__ == "__main__":
    main()
