"""Every schedule of the max-flow gives the oracle's masks.

The cut of an integer network is canonical, so who drives the rounds (host work lists or asynchronous single-launch
phases), how long a round is and whether the flow of the previous GrabCut iteration is kept must not change a single
pixel.  The switches are the documented GGC_MF_* variables of include/ggc.h, which the library reads ONCE per process,
so each variant runs in its own interpreter (one at a time: a GPU box admits few processes on its card)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import sys
sys.path[:0] = [r"{root}", r"{root}/src", r"{root}/tests"]
import numpy as np, torch
from gcn_grabcut import _native
from gcn_grabcut.synthetic import synthetic_image
from oracle import oracle as orc
ctx = _native.get_context(0)
h, w, b, n_iter = 200, 272, 3, 3                      # several 32x32 / 32x8 tiles each way, ragged right and bottom edges
pairs = [synthetic_image(h, w, 8100 + i, return_mask=True) for i in range(b)]
imgs = np.stack([p[0] for p in pairs])
tris = []
for img, gt in pairs:
    t = np.full((h, w), 2, np.uint8); t[gt == 1] = 3
    t[:3] = 0; t[-3:] = 0; t[:, :3] = 0; t[:, -3:] = 0
    ys, xs = np.nonzero(gt); t[int(ys.mean()) - 2:int(ys.mean()) + 3, int(xs.mean()) - 2:int(xs.mean()) + 3] = 1
    tris.append(t)
tris = np.stack(tris)
dimg = torch.as_tensor(imgs).cuda(); dmask = torch.as_tensor(tris).cuda()
bgd = torch.zeros(b, 65, dtype=torch.float64, device="cuda"); fgd = torch.zeros_like(bgd)
binary = torch.empty(b, h, w, dtype=torch.uint8, device="cuda")
ctx.call("ggc_grabcut", torch.cuda.current_stream().cuda_stream, b, h, w, dimg.data_ptr(), dmask.data_ptr(), None,
         bgd.data_ptr(), fgd.data_ptr(), n_iter, 0, 5, binary.data_ptr())
got = dmask.cpu().numpy()
for i in range(b):
    wb, wm, *_ = orc.grabcut(imgs[i], tris[i], n_iter=n_iter, mode=0, seed=5 + i)
    assert np.array_equal(got[i], wm), (i, int((got[i] != wm).sum()))
print("variant ok")
"""

VARIANTS = {
    "async_default": {},
    "host_work_lists": {"GGC_MF_ASYNC": "0"},
    "exact_relabel_every_round": {"GGC_MF_PARTIAL_ROUNDS": "0", "GGC_MF_RELAX_DENSE": "2"},
    "partial_relabels_for_long": {"GGC_MF_PARTIAL_ROUNDS": "9", "GGC_MF_RELAX_DENSE": "1"},
    "cold_start_every_iteration": {"GGC_MF_WARM": "0"},
    "async_all_push_rounds": {"GGC_MF_ASYNC_PUSH_ACTIVE": "100000000", "GGC_MF_RELAX_DENSE": "1"},
    "async_tiles_32x16": {"GGC_MF_ASYNC_TILE": "16", "GGC_MF_ASYNC_SWEEPS": "16"},
    "async_tiles_32x32_short_chains": {"GGC_MF_ASYNC_TILE": "32", "GGC_MF_ASYNC_HOPS": "8"},
    "short_dense_rounds": {"GGC_MF_DENSE_LAUNCHES0": "3", "GGC_MF_DENSE_LAUNCHES": "2", "GGC_MF_DENSE_SWEEPS": "4", "GGC_MF_RELAX_DENSE": "4"},
}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_driver_variant_matches_oracle(name, oracle):
    env = {k: v for k, v in os.environ.items() if not k.startswith("GGC_MF")}
    env.update(VARIANTS[name])
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=str(ROOT))], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "variant ok" in r.stdout, f"{name}: rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
