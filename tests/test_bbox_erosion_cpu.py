"""segment_bbox's definite-foreground core (reference pipeline.py:366-370: cv2.erode of the filled box with a 30x30 kernel) is
computed in closed form on the host; checked here against scipy's binary erosion with the same structure, the same anchor
(scipy centres an even-sized structure at size // 2, cv2 at ksize / 2) and cv2.erode's default border (outside counts as set).
cv2 itself is absent: parity with cv2.erode is unpinned, this pins the closed form against an independent erosion."""
import numpy as np
import pytest
from scipy import ndimage

from gcn_grabcut.pipeline import eroded_box


@pytest.mark.parametrize("H,W,bbox", [
    (100, 100, (10, 10, 80, 80)),        # reference tests/test.py:450-458
    (120, 160, (0, 0, 160, 120)),        # the whole frame: nothing is eroded
    (120, 160, (0, 20, 70, 90)),         # touches the left edge
    (120, 160, (100, 60, 60, 60)),       # touches the right and bottom edges
    (120, 160, (40, 40, 29, 50)),        # narrower than the kernel: empty core
    (120, 160, (40, 40, 30, 30)),        # exactly the kernel: one pixel
    (64, 64, (-5, -5, 40, 40)),          # negative corner: numpy's slice [-5:35] of 64 rows is empty, like in the reference
    (64, 64, (-30, 10, 50, 40)),         # x = -30: columns [-30:20] -> [34:20], empty
    (64, 64, (-50, -50, 60, 60)),        # [-50:10] -> [14:10], empty
    (64, 64, (20, 30, 100, 100)),        # stop past the frame: cut at the edge, which is then not eroded
])
@pytest.mark.parametrize("ksize", [30, 5])
def test_eroded_box_equals_binary_erosion(H, W, bbox, ksize):
    x, y, w, h = bbox
    inner = np.zeros((H, W), bool)
    inner[y:y + h, x:x + w] = True          # the reference's fill (pipeline.py:368), numpy slice semantics included
    want = ndimage.binary_erosion(inner, structure=np.ones((ksize, ksize), bool), border_value=1)
    y0, y1, x0, x1 = eroded_box(H, W, bbox, ksize)
    got = np.zeros((H, W), bool)
    if y1 > y0 and x1 > x0:
        got[y0:y1, x0:x1] = True
    assert np.array_equal(got, want)
