"""GPU parity for C0-C6 (GrabCut), K0 (clean_mask), O0 (compose), R0 (IoU).
Capacities are integers and the cut is canonical, so the masks must equal the
CPU oracle's bit for bit (IoU == 1)."""
import numpy as np
import pytest
import torch

import gpu_helpers as gh

pytestmark = pytest.mark.gpu


def _grabcut(ctx, img, mask, n_iter=5, mode=0, rects=None, seed=0, bgd=None, fgd=None):
    b, h, w, _ = img.shape
    dimg = torch.as_tensor(np.ascontiguousarray(img)).cuda()
    dmask = torch.as_tensor(np.ascontiguousarray(mask)).cuda() if mask is not None else torch.zeros(b, h, w, dtype=torch.uint8, device="cuda")
    dbgd = torch.zeros(b, 65, dtype=torch.float64, device="cuda") if bgd is None else torch.as_tensor(bgd).cuda()
    dfgd = torch.zeros(b, 65, dtype=torch.float64, device="cuda") if fgd is None else torch.as_tensor(fgd).cuda()
    binary = torch.empty(b, h, w, dtype=torch.uint8, device="cuda")
    r = None if rects is None else np.ascontiguousarray(rects, dtype=np.int32)
    ctx.call("ggc_grabcut", gh.stream(), b, h, w, dimg.data_ptr(), dmask.data_ptr(), None if r is None else r.ctypes.data,
             dbgd.data_ptr(), dfgd.data_ptr(), n_iter, mode, seed, binary.data_ptr())
    return binary.cpu().numpy(), dmask.cpu().numpy(), dbgd.cpu().numpy(), dfgd.cpu().numpy()


def _trimaps(imgs, gts):
    tris = []
    for img, gt in zip(imgs, gts):
        h, w = gt.shape
        tri = np.full((h, w), 2, np.uint8)
        tri[gt == 1] = 3
        tri[:3] = 0; tri[-3:] = 0; tri[:, :3] = 0; tri[:, -3:] = 0
        ys, xs = np.nonzero(gt)
        cy, cx = int(ys.mean()), int(xs.mean())
        tri[cy - 2:cy + 3, cx - 2:cx + 3] = 1
        tris.append(tri)
    return np.stack(tris)


@pytest.mark.parametrize("h,w,b,n_iter", [(48, 64, 3, 1), (96, 128, 2, 3), (300, 400, 2, 5)])
def test_grabcut_mask_is_bit_exact(oracle, gpu_ctx, h, w, b, n_iter):
    from gcn_grabcut.synthetic import synthetic_image
    pairs = [synthetic_image(h, w, 7000 + i, return_mask=True) for i in range(b)]
    imgs = np.stack([p[0] for p in pairs]); gts = [p[1] for p in pairs]
    tris = _trimaps(imgs, gts)
    binary, mask, bgd, fgd = _grabcut(gpu_ctx, imgs, tris, n_iter=n_iter, seed=11)
    for i in range(b):
        wb, wm, wbgd, wfgd, rc = oracle.grabcut(imgs[i], tris[i], n_iter=n_iter, mode=0, seed=11 + i)
        assert rc == 0
        assert np.array_equal(mask[i], wm), (i, int((mask[i] != wm).sum()), oracle.iou(binary[i], wb))
        assert np.array_equal(binary[i], wb)
        assert np.array_equal(bgd[i], wbgd) and np.array_equal(fgd[i], wfgd)     # GMMs from exact integer sums
        assert oracle.iou(binary[i], gts[i]) > 0.5


def test_grabcut_reference_fixture_cases(oracle, gpu_ctx):
    """reference tests/test.py:31-58 on its uniform-noise fixture: bbox mode, trimap
    mode with only probable labels (promotion branch), degenerate trimap."""
    img = np.random.RandomState(42).randint(20, 220, (100, 100, 3), dtype=np.uint8)
    binary, mask, *_ = _grabcut(gpu_ctx, img[None], None, n_iter=1, mode=1, rects=[[10, 10, 80, 80]])
    wb, wm, *_ = oracle.grabcut(img, None, n_iter=1, mode=1, rect=(10, 10, 80, 80))
    assert binary.shape == (1, 100, 100) and set(np.unique(binary)) <= {0, 1}
    assert np.array_equal(mask[0], wm) and np.array_equal(binary[0], wb)
    tri = np.full((100, 100), 2, np.uint8); tri[30:70, 30:70] = 3
    binary, mask, *_ = _grabcut(gpu_ctx, img[None], tri[None], n_iter=1, mode=0)
    wb, wm, *_ = oracle.grabcut(img, tri, n_iter=1, mode=0)
    assert np.array_equal(mask[0], wm) and np.array_equal(binary[0], wb)
    deg = np.full((100, 100), 3, np.uint8)                  # single class: grabcut.py:135-140
    binary, mask, *_ = _grabcut(gpu_ctx, img[None], deg[None], n_iter=5, mode=0)
    assert binary.all() and (mask == 1).all()


def test_grabcut_eval_mode_and_mixed_batch(oracle, gpu_ctx):
    from gcn_grabcut.synthetic import synthetic_image
    img, gt = synthetic_image(64, 80, 99, return_mask=True)
    tri = _trimaps([img], [gt])[0]
    deg = np.full_like(tri, 2)                               # second image of the batch is degenerate
    imgs, tris = np.stack([img, img]), np.stack([tri, deg])
    b1, m1, bgd, fgd = _grabcut(gpu_ctx, imgs, tris, n_iter=2, seed=5)
    assert (m1[1] == 0).all() and b1[1].sum() == 0
    b2, m2, _, _ = _grabcut(gpu_ctx, imgs[:1], m1[:1], n_iter=1, mode=2, bgd=bgd[:1], fgd=fgd[:1])
    b3, m3, _, _ = _grabcut(gpu_ctx, imgs[:1], tris[:1], n_iter=3, seed=5)
    assert np.array_equal(m2, m3)                            # 2 iterations + 1 continued == 3 iterations
    wb, wm, *_ = oracle.grabcut(img, tri, n_iter=3, mode=0, seed=5)
    assert np.array_equal(m3[0], wm)


def test_grabcut_rejects_bad_mask_values(gpu_ctx):
    from gcn_grabcut import _native
    img = np.zeros((1, 16, 16, 3), np.uint8)
    bad = np.full((1, 16, 16), 7, np.uint8)
    with pytest.raises(_native.GGCError, match="outside"):
        _grabcut(gpu_ctx, img, bad)


def test_clean_mask_compose_iou_bit_exact(oracle, gpu_ctx):
    rng = np.random.default_rng(0)
    h, w, b = 60, 70, 4
    masks = (rng.random((b, h, w)) < 0.08).astype(np.uint8)
    masks[0, 10:40, 10:50] = 1
    masks[1] = 0                                              # empty mask: returned as is
    masks[2, 5:8, 5:8] = 1; masks[2, 30:33, 30:33] = 1        # equal-area components: first in raster order wins
    masks[2] &= 0; masks[2, 5:8, 5:8] = 1; masks[2, 30:33, 30:33] = 1
    dm = torch.as_tensor(masks).cuda()
    out = torch.empty_like(dm)
    for ratio, keep in ((0.002, 0), (0.01, 0), (0.9, 0), (0.0, 1), (0.0, 0), (0.002, 1)):
        gpu_ctx.call("ggc_clean_mask", gh.stream(), b, h, w, dm.data_ptr(), ratio, keep, out.data_ptr())
        got = out.cpu().numpy()
        for i in range(b):
            assert np.array_equal(got[i], oracle.clean_mask(masks[i], ratio, bool(keep))), (ratio, keep, i)
    img = rng.integers(0, 256, (b, h, w, 3), dtype=np.uint8)
    di = torch.as_tensor(img).cuda()
    ov = torch.empty(b, h, w, 3, dtype=torch.uint8, device="cuda")
    rgba = torch.empty(b, h, w, 4, dtype=torch.uint8, device="cuda")
    gpu_ctx.call("ggc_compose_outputs", gh.stream(), b, h, w, di.data_ptr(), dm.data_ptr(), 0.45, 100, 220, 0,
                 ov.data_ptr(), rgba.data_ptr())
    iou = torch.empty(b, dtype=torch.float64, device="cuda")
    other = torch.as_tensor(np.roll(masks, 3, axis=2)).cuda()
    cnt = torch.empty(b, 3, dtype=torch.int64, device="cuda")
    gpu_ctx.call("ggc_mask_iou", gh.stream(), b, h, w, dm.data_ptr(), other.data_ptr(), iou.data_ptr(), cnt.data_ptr())
    for i in range(b):
        wo, wr = oracle.compose(img[i], masks[i])
        assert np.array_equal(ov[i].cpu().numpy(), wo) and np.array_equal(rgba[i].cpu().numpy(), wr)
        o = np.roll(masks, 3, axis=2)[i]
        assert iou[i].item() == pytest.approx(oracle.iou(masks[i], o), abs=1e-12)
        assert cnt[i].tolist() == [int((masks[i] & o).sum()), int((masks[i] & (1 - o)).sum()), int(((1 - masks[i]) & o).sum())]


def test_grabcut_lanes_equal_single_stream(gpu_ctx):
    """The pipeline runs GrabCut as concurrent sub-batches on private contexts / streams (Engine.grabcut_lanes);
    image b keeps seed + b, so masks and models must equal the single-stream call bit for bit."""
    from gcn_grabcut._engine import get_engine
    from gcn_grabcut.synthetic import synthetic_image
    eng = get_engine("cuda")
    pairs = [synthetic_image(96, 128, 7100 + i, return_mask=True) for i in range(10)]
    imgs = np.ascontiguousarray(np.stack([p[0] for p in pairs]))
    tri = _trimaps(imgs, [p[1] for p in pairs])
    bgr = torch.from_numpy(imgs).cuda()
    m1, m2 = torch.from_numpy(tri).cuda(), torch.from_numpy(tri).cuda()
    b1, m1, bg1, fg1 = eng.grabcut(bgr, m1, 3, 0, None, 5)
    b2, m2, bg2, fg2 = eng.grabcut_lanes(bgr, m2, 3, 0, 5, 4)
    torch.cuda.synchronize()
    assert torch.equal(b1, b2) and torch.equal(m1, m2)
    assert torch.equal(bg1, bg2) and torch.equal(fg1, fg2)
