"""Seeded synthetic inputs shared by the tests (test infrastructure)."""
import numpy as np
import torch


def chain_graph(n=80, seed=None, in_dim=19, edge_dim=5):
    """Chain graph with random features — mirrors reference tests/test.py:257-272."""
    gen = torch.Generator()
    gen.manual_seed(0 if seed is None else seed)
    x = torch.randn(n, in_dim, generator=gen)
    src = torch.arange(n - 1)
    dst = torch.arange(1, n)
    edge_index = torch.stack([torch.cat([src, dst]), torch.cat([dst, src])])
    edge_attr = torch.rand(edge_index.size(1), edge_dim, generator=gen)
    return x, edge_index, edge_attr


def superpixel_like_graph(n=600, k_nl=4, seed=0):
    """Jittered-grid region graph shaped like a DUTS superpixel graph:
    4-neighbour adjacency plus k nearest non-adjacent 'colour' neighbours,
    mirrored exactly like reference graph_builder.py:303-306. N~600 -> E~6.4k."""
    rng = np.random.default_rng(seed)
    gw = int(round(np.sqrt(n * 4 / 3)))
    gh = int(np.ceil(n / gw))
    ids = np.arange(gh * gw).reshape(gh, gw)
    keep = ids < n
    pairs = set()
    for a, b in ((ids[:, :-1], ids[:, 1:]), (ids[:-1, :], ids[1:, :]), (ids[:-1, :-1], ids[1:, 1:])):
        ok = (a < n) & (b < n)
        if a is ids[:-1, :-1]:
            ok &= rng.random(a.shape) < 0.35
        for u, v in zip(a[ok].ravel(), b[ok].ravel()):
            pairs.add((int(min(u, v)), int(max(u, v))))
    adj = sorted(pairs)
    col = rng.random((n, 3)).astype(np.float32)
    d = np.linalg.norm(col[:, None] - col[None], axis=2)
    np.fill_diagonal(d, np.inf)
    for u, v in adj:
        d[u, v] = d[v, u] = np.inf
    nb = np.argsort(d, axis=1, kind="stable")[:, :k_nl]
    nl = sorted({(int(min(i, j)), int(max(i, j))) for i in range(n) for j in nb[i]})
    pr = np.array(adj + nl, dtype=np.int64)
    src = np.concatenate([pr[:, 0], pr[:, 1]])
    dst = np.concatenate([pr[:, 1], pr[:, 0]])
    attr = rng.random((len(pr), 5)).astype(np.float32)
    attr[: len(adj), 4] = 0.0
    attr[len(adj):, 4] = 1.0
    attr[len(adj):, 2] = 0.0
    x = rng.random((n, 19)).astype(np.float32)
    return x, np.stack([src, dst]), np.concatenate([attr, attr], 0)


def seeded_state_dict(hidden=128, n_layers=6, seed=0, perturb=True):
    """Deterministic ResGCNNet weights: the reference init (model.py:501-506)
    under torch.manual_seed(seed), with biases / norms / BN stats / jk logits
    perturbed so that every term of the forward pass is exercised."""
    from gcn_grabcut.model import ResGCNNet
    torch.manual_seed(seed)
    m = ResGCNNet(hidden_channels=hidden, n_layers=n_layers)
    sd = m.state_dict()
    if perturb:
        g = torch.Generator()
        g.manual_seed(seed + 1)
        for k, v in sd.items():
            if not v.dtype.is_floating_point:
                continue
            if k.endswith("running_var"):
                v.copy_(0.5 + torch.rand(v.shape, generator=g))
            elif k.endswith("running_mean"):
                v.copy_(0.5 + 0.2 * torch.randn(v.shape, generator=g))
            elif v.dim() == 1:
                v.add_(0.1 * torch.randn(v.shape, generator=g))
        m.load_state_dict(sd)
    return m, {k: v.clone() for k, v in m.state_dict().items()}
