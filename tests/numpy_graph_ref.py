"""
Independent numpy formulation of the superpixel-graph stage, used to pin the C
oracle (oracle/graph.c).  It restates what reference graph_builder.py:190-454
computes with the numpy primitives SURVEY.md section 8(c) lists as in-container
oracles (bincount with weights, unique with counts, argpartition).
Test infrastructure only.
"""
import numpy as np

F = np.float32


def _groupsum(ids, values, n):
    return np.bincount(ids, weights=values.ravel(), minlength=n).astype(F)


def region_stats(seg, lab, hsv, grad, boundaries):
    h, w = seg.shape
    ids = seg.ravel()
    n = int(ids.max()) + 1
    cnt = np.bincount(ids, minlength=n).astype(F)
    den = np.maximum(cnt, F(1))
    mlab = np.stack([_groupsum(ids, lab[..., c], n) for c in range(3)], 1) / den[:, None]
    sqlab = np.stack([_groupsum(ids, lab[..., c] ** 2, n) for c in range(3)], 1) / den[:, None]
    mhsv = np.stack([_groupsum(ids, hsv[..., c], n) for c in range(3)], 1) / den[:, None]
    rows, cols = np.mgrid[0:h, 0:w]
    cy = _groupsum(ids, rows.astype(F) / h, n) / den
    cx = _groupsum(ids, cols.astype(F) / w, n) / den
    gn = grad / (grad.max() + 1e-6)
    return dict(n=n, cnt=cnt, den=den, mlab=mlab.astype(F),
                slab=np.sqrt(np.maximum(sqlab - mlab ** 2, 0.0)).astype(F), mhsv=mhsv.astype(F),
                cen=np.stack([cy, cx], 1).astype(F), bpx=_groupsum(ids, boundaries.astype(F), n),
                mgrad=(_groupsum(ids, grad, n) / den).astype(F), mgn=(_groupsum(ids, gn, n) / den).astype(F),
                area=(cnt / float(h * w)).astype(F))


def node_features(st):
    x = np.zeros((st["n"], 16), F)
    x[:, 0:3], x[:, 3:6], x[:, 6:9] = st["mlab"], st["slab"], st["mhsv"]
    x[:, 9:11] = st["cen"]
    x[:, 11] = st["area"]
    per = np.maximum(st["bpx"], 1.0)
    x[:, 12] = np.clip((4 * np.pi * st["cnt"]) / (per ** 2), 0.0, 1.0)
    x[:, 13] = st["mgrad"] / 255.0
    x[:, 14] = st["bpx"] / st["den"]
    x[:, 15] = np.linalg.norm(st["cen"] - 0.5, axis=1) / 0.707
    for sl in (slice(0, 3), slice(3, 6)):
        blk = x[:, sl]
        lo, hi = blk.min(0), blk.max(0)
        x[:, sl] = (blk - lo) / (hi - lo + 1e-6)
    return np.nan_to_num(x, nan=0.0, posinf=1.0, neginf=0.0)


def _pair_attr(pairs, st, shared, flag):
    i, j = pairs[:, 0], pairs[:, 1]
    de = np.linalg.norm(st["mlab"][i] - st["mlab"][j], axis=1)
    de = de / (de.max() + 1e-6)
    dc = np.linalg.norm(st["cen"][i] - st["cen"][j], axis=1)
    dc = dc / (dc.max() + 1e-6)
    gc = np.abs(st["mgn"][i] - st["mgn"][j])
    return np.stack([de, dc, shared, gc, np.full(len(pairs), flag, F)], 1).astype(F)


def edges(seg, st, connectivity=4, k=4):
    n = st["n"]
    views = [(seg[:, :-1], seg[:, 1:]), (seg[:-1], seg[1:])]
    if connectivity == 8:
        views += [(seg[:-1, :-1], seg[1:, 1:]), (seg[:-1, 1:], seg[1:, :-1])]
    a = np.concatenate([v[0].ravel() for v in views]).astype(np.int64)
    b = np.concatenate([v[1].ravel() for v in views]).astype(np.int64)
    m = a != b
    code, cnt = np.unique(np.minimum(a[m], b[m]) * n + np.maximum(a[m], b[m]), return_counts=True)
    pairs = np.stack([code // n, code % n], 1)
    shared = cnt.astype(F) / (cnt.max() + 1e-6)
    attr = _pair_attr(pairs, st, shared, 0.0)
    if k > 0 and n > k + 1:
        d = np.linalg.norm(st["mlab"][:, None] - st["mlab"][None], axis=2)
        np.fill_diagonal(d, np.inf)
        d[pairs[:, 0], pairs[:, 1]] = np.inf
        d[pairs[:, 1], pairs[:, 0]] = np.inf
        nb = np.argsort(d, axis=1, kind="stable")[:, :k]          # k smallest, ties -> lowest index
        r = np.repeat(np.arange(n), k)
        c = nb.ravel()
        ok = np.isfinite(d[r, c])
        r, c = r[ok], c[ok]
        nlc = np.unique(np.minimum(r, c).astype(np.int64) * n + np.maximum(r, c))
        nl = np.stack([nlc // n, nlc % n], 1)
        if len(nl):
            pairs = np.concatenate([pairs, nl])
            attr = np.concatenate([attr, _pair_attr(nl, st, np.zeros(len(nl), F), 1.0)])
    ei = np.stack([np.concatenate([pairs[:, 0], pairs[:, 1]]), np.concatenate([pairs[:, 1], pairs[:, 0]])])
    return ei, np.concatenate([attr, attr])


def _unit(v):
    v = v.astype(F)
    lo, hi = float(v.min()), float(v.max())
    return np.zeros_like(v) if hi - lo < 1e-8 else (v - lo) / (hi - lo)


def auto_prior(seg, lab, centre_sigma=0.45, contrast_sigma=0.40):
    h, w = seg.shape
    ids = seg.ravel()
    n = int(ids.max()) + 1
    cnt = np.bincount(ids, minlength=n).astype(F)
    den = np.maximum(cnt, F(1))
    mlab = np.stack([np.bincount(ids, weights=lab[..., c].ravel(), minlength=n) for c in range(3)], 1).astype(F) / den[:, None]
    rows, cols = np.mgrid[0:h, 0:w]
    cen = np.stack([np.bincount(ids, weights=rows.ravel() / h, minlength=n) / den,
                    np.bincount(ids, weights=cols.ravel() / w, minlength=n) / den], 1).astype(F)
    cd = np.linalg.norm(mlab[:, None] - mlab[None], axis=2)
    sd = np.linalg.norm(cen[:, None] - cen[None], axis=2)
    contrast = _unit((cd * np.exp(-(sd ** 2) / (2 * contrast_sigma ** 2)) * (cnt / max(cnt.sum(), 1.0))[None]).sum(1))
    cw = np.exp(-(np.linalg.norm(cen - 0.5, axis=1) ** 2) / (2 * centre_sigma ** 2))
    fg = _unit(contrast * cw)
    frame = np.concatenate([seg[0], seg[-1], seg[:, 0], seg[:, -1]])
    bc = np.bincount(frame, minlength=n).astype(F)
    if bc.sum() > 0:
        wb = bc / bc.sum()
        mu = (mlab * wb[:, None]).sum(0)
        var = (((mlab - mu) ** 2) * wb[:, None]).sum(0).sum()
        sg = float(np.sqrt(max(var, 1e-6)))
        bg = np.exp(-(np.linalg.norm(mlab - mu, axis=1) ** 2) / (2 * (sg + 1e-6) ** 2))
    else:
        bg = np.zeros(n, F)
    bg = _unit(np.maximum(bg, np.clip(bc / den * 4.0, 0.0, 1.0)))
    return np.nan_to_num(np.stack([fg, bg, 1.0 - np.abs(fg - bg)], 1).astype(F), nan=0.0, posinf=1.0, neginf=0.0)
