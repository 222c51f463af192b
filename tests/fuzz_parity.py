"""Parity fuzzer (a checker, hence under tests/; tests/test_fuzz_gpu.py runs a short fixed-seed pass of it): random image sizes, superpixel counts, batch sizes and image kinds through
GCNGrabCutPipeline.segment_batch_device against the CPU oracle (label map, trimap, mask of the first, middle and last image).
Environment: FUZZ_N cases (40), FUZZ_SEED (1), FUZZ_MIN / FUZZ_MAX image side (20 / 260), FUZZ_PX_PER_SEG (30), FUZZ_OPTIONS=1 also draws
the SuperpixelGraphConfig / segment() options (use_lab, connectivity, n_nonlocal, compactness, sigma, thresholds, refine_iters, ...).
    gpurun -- 'FUZZ_OPTIONS=1 FUZZ_N=120 python3 tests/fuzz_parity.py'
FUZZ_KIND=grabcut fuzzes the GrabCut class instead (trimap / rectangle starts, refine, colour spaces, 1-5 iterations).
Round 3: 60 + 150 (sides 6-70) + 25 (sides 250-640) + 120 (with options) pipeline cases and 150 GrabCut-class cases, no mismatch."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "src")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import seeded_state_dict
from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig
from gcn_grabcut.synthetic import synthetic_batch
from oracle import oracle as orc          # checker

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))


def fuzz_grabcut_class():
    """FUZZ_KIND=grabcut: the GrabCut class (run_with_trimap / run_with_bbox, then refine) against oracle.grabcut, incl. one-sided
    and empty trimaps, rectangles that touch or leave the image, 1-5 iterations, the three colour spaces."""
    from gcn_grabcut.grabcut import GrabCut, GrabCutConfig
    from gcn_grabcut.synthetic import synthetic_image
    bad = 0
    for case in range(int(os.environ.get("FUZZ_N", "40"))):
        h, w = int(rng.integers(8, 200)), int(rng.integers(8, 200))
        img = synthetic_image(h, w, int(rng.integers(0, 1000))) if rng.random() < 0.6 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        n_iter, cs, seed = int(rng.integers(1, 6)), str(rng.choice(["rgb", "rgb", "hsv", "lab"])), int(rng.integers(0, 100))
        gc = GrabCut(img, GrabCutConfig(n_iter=n_iter, color_space=cs, seed=seed))
        conv = img if cs == "rgb" else orc.convert_color8(img, cs)
        desc = f"grabcut case {case}: {h}x{w} it={n_iter} {cs} seed={seed}"
        if rng.random() < 0.5:
            x0, y0 = int(rng.integers(-3, w - 1)), int(rng.integers(-3, h - 1))
            rect = (x0, y0, int(rng.integers(1, w + 4)), int(rng.integers(1, h + 4)))
            desc += f" rect={rect}"
            try:
                got = gc.run_with_bbox(rect)
            except Exception as e:
                print(desc, "raised", f"{type(e).__name__}: {e}"[:200]); continue
            wb, wm, bgd, fgd, rc = orc.grabcut(conv, None, n_iter=n_iter, mode=1, rect=rect, seed=seed)
        else:
            tri = np.full((h, w), int(rng.choice([2, 3])), np.uint8)
            for _ in range(int(rng.integers(0, 5))):                          # a few rectangles of random labels (0..3), maybe none
                ya, xa = int(rng.integers(0, h)), int(rng.integers(0, w))
                tri[ya:ya + int(rng.integers(1, h)), xa:xa + int(rng.integers(1, w))] = int(rng.integers(0, 4))
            desc += f" trimap labels={np.unique(tri).tolist()}"
            got = gc.run_with_trimap(tri)
            wb, wm, bgd, fgd, rc = orc.grabcut(conv, tri, n_iter, 0, None, seed)
        ok = np.array_equal(got, wb) and np.array_equal(gc.mask, wm)
        if ok and rc == 0 and rng.random() < 0.5:
            k = int(rng.integers(1, 4))
            got2 = gc.refine(k)
            wb2, wm2, *_ = orc.grabcut(conv, wm, k, 2, None, seed, bgd, fgd)
            ok = np.array_equal(got2, wb2) and np.array_equal(gc.mask, wm2)
            desc += f" refine={k}"
        if not ok: print(desc, "MISMATCH"); bad += 1
        elif case % 10 == 0: print(desc, "ok", flush=True)
    print("FUZZ DONE: mismatching cases", bad)


if os.environ.get("FUZZ_KIND", "pipeline") == "grabcut":
    fuzz_grabcut_class()
    sys.exit(0)

model, sd = seeded_state_dict(32, 2, seed=5)
st = {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
bad = 0
t0 = time.time()
for case in range(int(os.environ.get("FUZZ_N", "40"))):
    lo, hi = int(os.environ.get("FUZZ_MIN", "20")), int(os.environ.get("FUZZ_MAX", "260"))
    h, w = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
    n_seg = int(rng.integers(1, max(3, min(400, h * w // int(os.environ.get("FUZZ_PX_PER_SEG", "30"))))))
    b = int(rng.choice([1, 2, 3, 5, 9, 33]))
    kind = int(rng.integers(0, 4))
    if kind == 0: imgs = synthetic_batch(b, h, w, config_id=int(rng.integers(0, 9)))
    elif kind == 1: imgs = rng.integers(0, 256, (b, h, w, 3), dtype=np.uint8)                       # noise
    elif kind == 2: imgs = np.full((b, h, w, 3), int(rng.integers(0, 256)), np.uint8)               # flat
    else:                                                                                           # two flat halves + a little noise
        imgs = np.zeros((b, h, w, 3), np.uint8); imgs[:, :, w // 2:] = 200
        imgs = np.clip(imgs.astype(int) + rng.integers(-3, 4, imgs.shape), 0, 255).astype(np.uint8)
    opt = {}
    if os.environ.get("FUZZ_OPTIONS", "0") == "1":                        # options of the pipeline, each off its default now and then
        opt = dict(use_lab=bool(rng.random() < 0.7), connectivity=int(rng.choice([4, 8])), n_nonlocal=int(rng.choice([0, 2, 4, 6])),
                   compactness=float(rng.choice([1.0, 10.0, 30.0])), sigma=float(rng.choice([0.0, 1.0, 2.0])),
                   threshold_fg=float(rng.choice([0.4, 0.55, 0.8])), threshold_bg=float(rng.choice([0.4, 0.55, 0.8])),
                   refine_iters=int(rng.choice([0, 0, 2])), keep_largest=bool(rng.random() < 0.3), edge_aware=bool(rng.random() < 0.7),
                   filter_radius=int(rng.choice([2, 8, 12])), min_area_ratio=float(rng.choice([0.0, 0.002, 0.05])))
    sp = {k: opt[k] for k in ("use_lab", "connectivity", "n_nonlocal", "compactness", "sigma") if k in opt}
    run = {k: opt[k] for k in ("threshold_fg", "threshold_bg", "refine_iters", "keep_largest", "edge_aware", "filter_radius", "min_area_ratio") if k in opt}
    desc = f"case {case}: b={b} {h}x{w} n_seg={n_seg} kind={kind} {opt}"
    try:
        pipe = GCNGrabCutPipeline(model.eval(), sp_config=SuperpixelGraphConfig(n_segments=n_seg, **sp), device="cuda")
        out = pipe.segment_batch_device(torch.from_numpy(imgs).cuda(), **run)
        torch.cuda.synchronize()
    except Exception as e:
        msg = f"{type(e).__name__}: {e}"
        try:
            want = orc.segment(imgs[0], st, 32, 2, n_segments=n_seg, seed=0, **opt)
            print(desc, "PRODUCT RAISED but the oracle ran:", msg[:300]); bad += 1
        except Exception as e2:
            print(desc, "both raise:", msg[:120], "|", f"{type(e2).__name__}: {e2}"[:120])
        continue
    seg = out["segments"].cpu().numpy(); tri = out["trimap"].cpu().numpy(); bm = out["binary_mask"].cpu().numpy()
    ok = True
    for i in sorted(set([0, b - 1, b // 2])):
        want = orc.segment(imgs[i], st, 32, 2, n_segments=n_seg, seed=i, **opt)
        for name, got, exp in (("segments", seg[i], want["segments"]), ("trimap", tri[i], want["trimap"]), ("mask", bm[i], want["binary_mask"])):
            if not np.array_equal(got, exp):
                print(desc, f"image {i}: {name} differs in {(got != exp).sum()} px"); ok = False; break
        if not ok: break
    bad += 0 if ok else 1
    if ok and case % 5 == 0: print(desc, "ok", f"({time.time() - t0:.0f} s)", flush=True)
print("FUZZ DONE: mismatching cases", bad)
