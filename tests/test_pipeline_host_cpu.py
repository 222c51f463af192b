"""Host logic of the software pipeline (no GPU): the chunk plan and the merge of per-chunk graphs into one packed batch."""
import numpy as np
import pytest
import torch


def test_chunk_plan_covers_the_batch_in_order():
    from gcn_grabcut.pipeline import GCNGrabCutPipeline as P
    for b, n, ratio in ((256, 4, 1.0), (256, 4, 0.7), (256, 8, 0.85), (22, 3, 1.0), (5, 9, 0.8), (1, 4, 0.5), (17, 2, 0.1)):
        plan = P.chunk_plan(b, n, ratio)
        assert plan[0][0] == 0 and plan[-1][1] == b and len(plan) == min(n, b)
        assert all(lo < hi for lo, hi in plan)                               # no empty chunk
        assert all(plan[k][1] == plan[k + 1][0] for k in range(len(plan) - 1))   # contiguous, in order
    assert P.chunk_plan(256, 4, 1.0) == [(0, 64), (64, 128), (128, 192), (192, 256)]
    sizes = [hi - lo for lo, hi in P.chunk_plan(256, 4, 0.7)]
    assert sizes == sorted(sizes, reverse=True) and sizes[0] > 64 > sizes[-1]  # later chunks start later: fewer images


def test_merge_graphs_is_the_packed_batch():
    from gcn_grabcut._engine import DeviceGraphs, merge_graphs
    rng = np.random.default_rng(0)

    def part(n_nodes_per_image, e_per_image):
        node_ptr = np.concatenate([[0], np.cumsum(n_nodes_per_image)]).astype(np.int64)
        edge_ptr = np.concatenate([[0], np.cumsum(e_per_image)]).astype(np.int64)
        n, e = int(node_ptr[-1]), int(edge_ptr[-1])
        src = np.concatenate([rng.integers(node_ptr[i], node_ptr[i + 1], e_per_image[i]) for i in range(len(e_per_image))]).astype(np.int32)
        dst = np.concatenate([rng.integers(node_ptr[i], node_ptr[i + 1], e_per_image[i]) for i in range(len(e_per_image))]).astype(np.int32)
        return DeviceGraphs(torch.zeros(len(n_nodes_per_image), 2, 2, dtype=torch.int32), torch.tensor(n_nodes_per_image, dtype=torch.int32),
                            node_ptr, edge_ptr, torch.from_numpy(node_ptr.astype(np.int32)), torch.from_numpy(rng.random((n, 19), dtype=np.float32)),
                            torch.from_numpy(rng.random((n, 2), dtype=np.float32)), torch.from_numpy(rng.random(n, dtype=np.float32)),
                            torch.from_numpy(src), torch.from_numpy(dst), torch.from_numpy(rng.random((e, 5), dtype=np.float32)))

    parts = [part([5, 7], [10, 12]), part([3], [4]), part([6, 2, 4], [8, 0, 6])]
    seg = torch.zeros(6, 2, 2, dtype=torch.int32)
    g = merge_graphs(parts, seg)
    assert g.node_ptr_host.tolist() == [0, 5, 12, 15, 21, 23, 27] and g.edge_ptr_host.tolist() == [0, 10, 22, 26, 34, 34, 40]
    assert g.node_ptr.tolist() == g.node_ptr_host.tolist() and g.n_nodes.tolist() == [5, 7, 3, 6, 2, 4]
    assert g.x.shape == (27, 19) and torch.equal(g.x[12:15], parts[1].x) and torch.equal(g.edge_attr[22:26], parts[1].edge_attr)
    # edge endpoints are shifted by the nodes that precede their chunk and stay inside their own image
    src, dst = g.edge_src.numpy(), g.edge_dst.numpy()
    for i in range(6):
        e0, e1, n0, n1 = g.edge_ptr_host[i], g.edge_ptr_host[i + 1], g.node_ptr_host[i], g.node_ptr_host[i + 1]
        assert ((src[e0:e1] >= n0) & (src[e0:e1] < n1)).all() and ((dst[e0:e1] >= n0) & (dst[e0:e1] < n1)).all()
    assert np.array_equal(src[22:26] - 12, parts[1].edge_src.numpy()) and np.array_equal(dst[26:34] - 15, parts[2].edge_dst.numpy()[:8])
    assert merge_graphs(parts[:1], seg) is parts[0]
