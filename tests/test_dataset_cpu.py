"""Host logic of the graph-cache writer: descriptors, cache keys, label derivation."""
import hashlib
import zlib

import numpy as np
import pytest

from gcn_grabcut import dataset as ds
from gcn_grabcut.graph_builder import SuperpixelGraphConfig


def _write(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path)


def test_list_pairs_orders_and_seeds(tmp_path):
    (tmp_path / "im").mkdir(); (tmp_path / "mk").mkdir()
    rng = np.random.default_rng(0)
    for name in ("b", "a", "c"):
        _write(tmp_path / "im" / f"{name}.jpg", rng.integers(0, 255, (8, 9, 3), dtype=np.uint8))
    _write(tmp_path / "mk" / "a.png", np.zeros((8, 9), np.uint8))
    _write(tmp_path / "mk" / "c.bmp", np.zeros((8, 9), np.uint8))
    out = ds.list_image_mask_pairs(tmp_path / "im", tmp_path / "mk", max_size=384, augment_copies=2, seed=42)
    assert [s["name"] for s in out] == ["a", "a_aug0", "a_aug1", "c", "c_aug0", "c_aug1"]      # b has no mask
    assert out[0]["aug_seed"] is None and out[0]["max_size"] == 384
    # reference dataset.py:303-309: seed + 1000003 * k + crc32(stem) % 100003
    assert out[2]["aug_seed"] == 42 + 1000003 * 1 + zlib.crc32(b"a") % 100003
    assert out[3]["mask_path"].endswith("c.bmp")


def test_cache_key_is_the_reference_recipe():
    s = {"image_path": "/d/im/x.jpg", "mask_path": "/d/mk/x.png", "max_size": 384, "aug_seed": None}
    cfg = SuperpixelGraphConfig(n_segments=300)
    h = hashlib.sha1()
    h.update(repr(("/d/im/x.jpg", "/d/mk/x.png", 384, None)).encode())
    h.update(repr((cfg.n_segments, cfg.compactness, cfg.sigma, cfg.use_lab, cfg.connectivity, cfg.n_nonlocal, 0.7, 0.7)).encode())
    assert ds._cache_key(s, cfg, 0.7, 0.7) == h.hexdigest()[:20]
    assert ds._cache_key(s, cfg, 0.7, 0.7) != ds._cache_key(s, SuperpixelGraphConfig(n_segments=301), 0.7, 0.7)
    mem = {"image": np.zeros((4, 4, 3), np.uint8), "gt_mask": np.zeros((4, 4), np.uint8)}
    assert len(ds._cache_key(mem, None, 0.7, 0.7)) == 20


def test_derive_trimap_labels_matches_the_reference_formula():
    rng = np.random.default_rng(1)
    seg = rng.integers(0, 40, (60, 70)).astype(np.int32)
    seg[seg == 7] = 8                                      # an empty region id
    gt = (rng.random((60, 70)) < 0.5).astype(np.uint8)
    gt[seg < 10] = 1; gt[(seg >= 10) & (seg < 20)] = 0
    got = ds.derive_trimap_labels(seg, gt, 0.75, 0.75)
    n = int(seg.max()) + 1
    counts = np.bincount(seg.ravel(), minlength=n).astype(np.float64)       # dataset.py:197-205 verbatim arithmetic
    ratio = np.bincount(seg.ravel(), weights=(gt.ravel() > 0).astype(np.float64), minlength=n) / np.maximum(counts, 1.0)
    want = np.full(n, 1, np.int64); want[ratio >= 0.75] = 2; want[ratio <= 0.25] = 0; want[counts == 0] = 1
    assert got.dtype == np.int64 and np.array_equal(got, want)
    assert got[7] == 1 and (got[:7] == 2).all() and (got[10:20] == 0).all()


def test_materialise_decodes_resizes_and_filters(tmp_path):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 255, (100, 160, 3), dtype=np.uint8)
    mask = np.zeros((100, 160), np.uint8); mask[20:80, 30:120] = 255
    _write(tmp_path / "i.png", img[:, :, ::-1]); _write(tmp_path / "m.png", mask)          # file holds RGB
    s = {"image_path": str(tmp_path / "i.png"), "mask_path": str(tmp_path / "m.png"), "max_size": 512, "name": "i", "aug_seed": None}
    m = ds.materialise(s)
    assert np.array_equal(m["image"], img) and m["gt_mask"].sum() == 60 * 90 and m["image"].flags.c_contiguous
    m = ds.materialise({**s, "max_size": 80})
    assert m["image"].shape == (50, 80, 3) and m["gt_mask"].shape == (50, 80) and set(np.unique(m["gt_mask"])) == {0, 1}
    assert ds.materialise({**s, "mask_path": str(tmp_path / "nope.png")}) is None
    _write(tmp_path / "m2.png", np.zeros((100, 160), np.uint8))
    assert ds.materialise({**s, "mask_path": str(tmp_path / "m2.png")}) is None            # degenerate: no foreground
    assert ds.materialise({"image": img, "gt_mask": mask}) is not None                       # in-memory samples pass through


def test_seeded_augmentation_follows_the_reference_draw_order(tmp_path):
    """reference dataset.py:107-168, 343-353: the decisions come from `random` seeded with aug_seed, in the order
    flip? rotate? [angle] colour? [brightness, contrast, saturation] crop? [scale, y0, x0]; the global RNG state is restored."""
    import random
    rng = np.random.default_rng(3)
    img = rng.integers(0, 255, (120, 160, 3), dtype=np.uint8)
    mask = np.zeros((120, 160), np.uint8); mask[30:90, 40:130] = 255
    _write(tmp_path / "i.png", img[:, :, ::-1]); _write(tmp_path / "m.png", mask)
    base = {"image_path": str(tmp_path / "i.png"), "mask_path": str(tmp_path / "m.png"), "max_size": 512, "name": "i"}
    random.seed(1234)
    before = random.getstate()
    a = ds.materialise({**base, "aug_seed": 77})
    b = ds.materialise({**base, "aug_seed": 77})
    c = ds.materialise({**base, "aug_seed": 78})
    assert random.getstate() == before                                     # caller's RNG untouched
    assert np.array_equal(a["image"], b["image"]) and np.array_equal(a["gt_mask"], b["gt_mask"])     # same seed, same copy
    assert not np.array_equal(a["image"], c["image"])
    assert a["image"].shape == img.shape and a["image"].dtype == np.uint8 and set(np.unique(a["gt_mask"])) <= {0, 1}
    # replay the draw sequence by hand for seed 77 and compare the geometric decisions
    r = random.Random(77)
    flip = r.random() < 0.5
    rot = r.random() < 0.4
    angle = r.uniform(-15, 15) if rot else None
    col = r.random() < 0.6
    if col:
        [r.uniform(-40, 40), r.uniform(0.7, 1.3), r.uniform(0.7, 1.3)]
    crop = r.random() < 0.4
    plain = ds.materialise({**base, "aug_seed": None})
    if not rot and not crop:                                               # pure flip (+ colour): the mask is the flipped mask
        want = plain["gt_mask"][:, ::-1] if flip else plain["gt_mask"]
        assert np.array_equal(a["gt_mask"], want)
    # the pieces, on their own
    assert np.array_equal(ds._warp_affine(img, ds._rotation_matrix(80, 60, 0.0), linear=True), img)      # identity warp
    m90 = ds._rotation_matrix(59.5, 59.5, 90.0)
    sq = rng.integers(0, 255, (120, 120), dtype=np.uint8)
    assert np.array_equal(ds._warp_affine(sq, m90, linear=False), np.rot90(sq))                          # cv2: positive angle = counter-clockwise
    for bgr, hsv in (((0, 0, 255), (0, 255, 255)), ((0, 255, 0), (60, 255, 255)), ((255, 0, 0), (120, 255, 255)), ((128, 128, 128), (0, 0, 128))):
        px = np.array([[bgr]], np.uint8)
        assert tuple(ds._bgr_to_hsv8(px)[0, 0]) == hsv and tuple(ds._hsv8_to_bgr(np.array([[hsv]], np.uint8))[0, 0]) == bgr
    back = ds._hsv8_to_bgr(ds._bgr_to_hsv8(img))
    assert np.abs(back.astype(int) - img.astype(int)).max() <= 4           # 8-bit round trip (H has 2-degree steps)


def test_seeded_augmentation_is_the_same_on_worker_threads(tmp_path):
    """prepare_dataset decodes on a thread pool (the reference uses worker PROCESSES, each with its own `random`): the seeded
    copy must not depend on which thread makes it or on what the other threads draw meanwhile — and random.Random(seed) is
    the stream random.seed(seed) gives, so the reference's parameter sequence is kept."""
    import random
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(5)
    img = rng.integers(0, 255, (96, 128, 3), dtype=np.uint8)
    mask = np.zeros((96, 128), np.uint8); mask[20:80, 30:100] = 255
    _write(tmp_path / "i.png", img[:, :, ::-1]); _write(tmp_path / "m.png", mask)
    descs = [{"image_path": str(tmp_path / "i.png"), "mask_path": str(tmp_path / "m.png"), "max_size": 512, "name": f"i{k}",
              "aug_seed": 1000 + k} for k in range(32)]
    serial = [ds.materialise(d) for d in descs]
    for _ in range(3):
        with ThreadPoolExecutor(4) as pool:
            threaded = list(pool.map(ds.materialise, descs))
        for a, b in zip(serial, threaded):
            assert (a is None) == (b is None)
            if a is not None:
                assert np.array_equal(a["image"], b["image"]) and np.array_equal(a["gt_mask"], b["gt_mask"])
    # the private generator draws what the seeded global one would: augment_sample with the module-level `random` after
    # random.seed(s) (the reference's way) gives the same copy
    plain = ds.materialise({**descs[0], "aug_seed": None})
    state = random.getstate()
    try:
        random.seed(1003)
        via_global = ds.augment_sample(plain["image"], plain["gt_mask"], prob_flip=0.5, prob_rotate=0.4, prob_color=0.6, prob_crop=0.4)
    finally:
        random.setstate(state)
    via_private = ds.augment_sample(plain["image"], plain["gt_mask"], prob_flip=0.5, prob_rotate=0.4, prob_color=0.6, prob_crop=0.4,
                                    rng=random.Random(1003))
    assert np.array_equal(via_global[0], via_private[0]) and np.array_equal(via_global[1], via_private[1])
