"""The command-line shell of the hot path (reference inference.py): flags :26-56, checkpoint width / depth recovery :81-86,
output file names :146-155, per-image timing line :157-162 — run as a child process on three synthetic PNGs with a saved
seeded checkpoint."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import seeded_state_dict

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _png(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path)


def _run(args, cwd):
    return subprocess.run([sys.executable, str(ROOT / "inference.py")] + args, cwd=cwd, capture_output=True, text=True, timeout=600)


def test_inference_cli_writes_the_reference_outputs(tmp_path, oracle):
    from gcn_grabcut.synthetic import synthetic_image
    in_dir, out_dir = tmp_path / "in", tmp_path / "out"
    in_dir.mkdir()
    imgs = {}
    for k, (h, w) in enumerate(((120, 160), (120, 160), (96, 128))):      # two shapes: two device batches
        imgs[f"im{k}"] = synthetic_image(h, w, 50_000 + k)
        _png(in_dir / f"im{k}.png", imgs[f"im{k}"][:, :, ::-1])              # files hold RGB
    (in_dir / "notes.txt").write_text("not an image")
    model, sd = seeded_state_dict(64, 3, seed=21)                            # width / depth differ from the CLI defaults (128, 6)
    ckpt = tmp_path / "ckpt.pt"
    torch.save({"model": sd, "epoch": 3}, ckpt)
    r = _run(["--input", str(in_dir), "--output", str(out_dir), "--checkpoint", str(ckpt), "--superpixels", "150",
              "--save", "mask", "overlay", "rgba", "trimap", "--max-size", "0"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "loaded ResGCNNet (D=64, n=3)" in r.stdout                        # recovered from the tensors, not from --hidden / --layers
    names = sorted(p.name for p in out_dir.iterdir())
    assert names == sorted(f"im{k}_{s}.png" for k in range(3) for s in ("mask", "overlay", "rgba", "trimap"))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("[") and "fg=" in ln]
    assert len(lines) == 3 and all(all(key in ln for key in ("graph=", "gcn=", "grabcut=", "total=")) for ln in lines)
    assert "3 image(s)" in r.stdout
    # the written mask is the pipeline's mask, which is the oracle's
    from PIL import Image
    st = {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
    for k, seed in ((0, 0), (1, 1), (2, 0)):                                 # position inside its shape batch = GrabCut seed offset
        mask = np.asarray(Image.open(out_dir / f"im{k}_mask.png"))
        want = oracle.segment(imgs[f"im{k}"], st, 64, 3, n_segments=150, seed=seed)
        assert mask.shape == imgs[f"im{k}"].shape[:2] and set(np.unique(mask)) <= {0, 255}
        assert np.array_equal(mask // 255, want["binary_mask"])
        rgba = np.asarray(Image.open(out_dir / f"im{k}_rgba.png"))
        assert rgba.shape == (*mask.shape, 4) and np.array_equal(rgba[:, :, 3], mask)


def test_inference_cli_single_image_and_errors(tmp_path):
    from gcn_grabcut.synthetic import synthetic_image
    img = synthetic_image(90, 120, 51_000)
    _png(tmp_path / "one.png", img[:, :, ::-1])
    model, sd = seeded_state_dict(32, 2, seed=5)
    torch.save({"model": sd}, tmp_path / "m.pt")
    r = _run(["--image", str(tmp_path / "one.png"), "--output", str(tmp_path / "res"), "--checkpoint", str(tmp_path / "m.pt"),
              "--keep-largest", "--no-edge-aware", "--refine", "1", "--save", "mask"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(p.name for p in (tmp_path / "res").iterdir()) == ["one_mask.png"]
    r = _run(["--image", str(tmp_path / "one.png"), "--checkpoint", str(tmp_path / "missing.pt")], tmp_path)
    assert r.returncode != 0 and "No checkpoint" in r.stderr
    r = _run(["--input", str(tmp_path / "empty_dir_that_does_not_exist"), "--checkpoint", str(tmp_path / "m.pt")], tmp_path)
    assert r.returncode != 0
