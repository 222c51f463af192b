#!/opt/conda/bin/python3.9
"""
Generates tests/golden/skimage_0183_rgb.npz: the float64 SLIC path that the reference takes with
SuperpixelGraphConfig(use_lab=False) (graph_builder.py:177-179: `slic(self.rgb.astype(float), ...)`), run on
scikit-image 0.18.3 + scipy 1.7.1 under /opt/conda/bin/python3.9 (build container only):

    /opt/conda/bin/python3.9 tests/golden/make_golden_skimage_rgb.py

A float64 input keeps every stage of skimage's slic in float64 (img_as_float leaves it alone, rgb2lab, gaussian_filter,
_slic_cython's fused-type double instance).  skimage >= 0.19 first rescales the input to [0, 1] by its global min / max;
0.18.3 has no such step, so it is applied here before the 0.18.3 body (as make_golden_skimage.py does for the Lab path).
The fixtures are data (inputs and library outputs); nothing of the reference repository is involved.
"""
import importlib.util
import os
import sys

import numpy as np
import scipy
import skimage
from scipy import ndimage as ndi
from skimage.color import rgb2lab
from skimage.segmentation import slic
from skimage.segmentation._slic import _enforce_label_connectivity_cython, _slic_cython
from skimage.util import regular_grid

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
spec = importlib.util.spec_from_file_location("synthetic", os.path.join(ROOT, "gcn-grabcut_amd", "gcn_grabcut", "synthetic.py"))
synthetic = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synthetic)

CASES = [(64, 64, 50, 1), (72, 96, 120, 2), (96, 128, 200, 3), (50, 81, 300, 4)]      # (H, W, n_segments, seed)


def case(h, w, n_segments, seed, compactness=10.0, sigma=1.0):
    bgr = synthetic.synthetic_image(h, w, seed)
    rgb = np.ascontiguousarray(bgr[:, :, ::-1])
    x = rgb.astype(float)                            # what the reference hands to slic
    x -= x.min()                                     # skimage >= 0.19: global min-max rescale
    imax = x.max()
    if imax != 0:
        x /= imax
    img = x[np.newaxis, ...]
    img_lab = rgb2lab(img)
    assert img_lab.dtype == np.float64
    slices = regular_grid(img_lab.shape[:3], n_segments)
    gz, gy, gx = np.mgrid[:1, :h, :w]
    cent = np.concatenate([g[slices].ravel()[..., None] for g in (gz, gy, gx)], axis=-1)
    steps = np.asarray([float(s.step) if s.step is not None else 1.0 for s in slices])
    smoothed = ndi.gaussian_filter(img_lab, [sigma, sigma, sigma, 0])
    assert smoothed.dtype == np.float64
    k = cent.shape[0]
    segments0 = np.ascontiguousarray(np.concatenate([cent, np.zeros((k, 3))], axis=-1), dtype=np.float64)
    step = float(max(steps))
    scaled = np.ascontiguousarray(smoothed * (1.0 / compactness), dtype=np.float64)
    seg_work = segments0.copy()
    raw = _slic_cython(scaled, None, seg_work, step, 10, np.ones(3, np.float64), False, ignore_color=False, start_label=0)
    seg_size = np.prod(scaled.shape[:3]) / k
    min_size, max_size = int(0.5 * seg_size), int(3 * seg_size)
    conn = _enforce_label_connectivity_cython(np.ascontiguousarray(raw), min_size, max_size, start_label=0)
    whole = slic(x, n_segments=n_segments, compactness=compactness, sigma=sigma, start_label=0, multichannel=True)
    assert np.array_equal(whole, conn[0])
    return dict(bgr=bgr, rescaled=x, second_lab=img_lab[0], smoothed=smoothed[0], scaled=scaled[0], step=np.float64(step),
                centers_final=seg_work, raw=raw[0].astype(np.int32), connected=conn[0].astype(np.int32),
                n_segments=np.int64(n_segments))


def main():
    out = {}
    for i, (h, w, n, seed) in enumerate(CASES):
        for key, val in case(h, w, n, seed).items():
            out[f"r{i}_{key}"] = val
    out["versions"] = np.array([skimage.__version__, scipy.__version__, np.__version__, sys.version.split()[0]])
    np.savez_compressed(os.path.join(HERE, "skimage_0183_rgb.npz"), **out)
    print("wrote", os.path.join(HERE, "skimage_0183_rgb.npz"), "cases:", len(CASES))


if __name__ == "__main__":
    main()
