"""
PyG-free torch-CPU restatement of ResGCNNet.forward (reference model.py:508-536)
used to pin the C oracle.  GCNConv / SAGEConv follow SURVEY.md Appendix A.3.
Test infrastructure only.
"""
import torch
import torch.nn.functional as F


def gcn_conv(x, edge_index, weight, bias):
    n = x.size(0)
    xw = x @ weight.t()
    loops = torch.arange(n, dtype=edge_index.dtype)
    src = torch.cat([edge_index[0], loops])
    dst = torch.cat([edge_index[1], loops])
    deg = torch.zeros(n, dtype=x.dtype).scatter_add_(0, dst, torch.ones(dst.numel(), dtype=x.dtype))
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0
    norm = dis[src] * dis[dst]
    out = torch.zeros_like(xw).index_add_(0, dst, norm.unsqueeze(1) * xw[src])
    return out + bias


def scatter_mean(src, index, n):
    out = torch.zeros(n, src.size(1), dtype=src.dtype)
    out.scatter_add_(0, index.unsqueeze(1).expand_as(src), src)
    cnt = torch.bincount(index, minlength=n).to(src.dtype).clamp(min=1)
    return out / cnt.unsqueeze(1)


def sage_conv(x, edge_index, w_l, b_l, w_r):
    m = scatter_mean(x[edge_index[0]], edge_index[1], x.size(0))
    return m @ w_l.t() + b_l + x @ w_r.t()


def graph_softmax(scores, batch):
    if batch is None:
        return torch.softmax(scores, dim=0)
    g = int(batch.max()) + 1
    peak = torch.full((g, 1), float("-inf")).index_reduce(0, batch, scores, "amax", include_self=True)
    ex = torch.exp(scores - peak[batch])
    tot = torch.zeros_like(peak).index_add_(0, batch, ex)
    return ex / (tot[batch] + 1e-12)


@torch.no_grad()
def resgcn_forward(sd, n_layers, x, edge_index, edge_attr, batch=None):
    """sd: state_dict of float32 CPU tensors with the reference keys."""
    n = x.size(0)
    d = sd["input_proj.0.weight"].size(0)
    c = sd["edge_ctx.encode.0.weight"].size(0)
    xn = F.batch_norm(x, sd["in_norm.norm.running_mean"], sd["in_norm.norm.running_var"],
                      sd["in_norm.norm.weight"], sd["in_norm.norm.bias"], False, 0.0, 1e-5)
    h = F.gelu(F.layer_norm(F.linear(xn, sd["input_proj.0.weight"], sd["input_proj.0.bias"]), (d,),
                            sd["input_proj.1.weight"], sd["input_proj.1.bias"]))
    prior = x[:, -3:]
    pb = torch.sigmoid(F.linear(F.gelu(F.linear(prior, sd["prior_booster.0.weight"], sd["prior_booster.0.bias"])),
                                sd["prior_booster.2.weight"], sd["prior_booster.2.bias"]))
    h = h * (1.0 + pb)
    enc = F.linear(F.gelu(F.linear(edge_attr, sd["edge_ctx.encode.0.weight"], sd["edge_ctx.encode.0.bias"])),
                   sd["edge_ctx.encode.2.weight"], sd["edge_ctx.encode.2.bias"])
    ctx = scatter_mean(enc, edge_index[1], n)
    gate = torch.sigmoid(F.linear(F.layer_norm(ctx, (c,), sd["edge_ctx.to_gate.0.weight"], sd["edge_ctx.to_gate.0.bias"]),
                                  sd["edge_ctx.to_gate.1.weight"], sd["edge_ctx.to_gate.1.bias"]))
    states = [h]
    for i in range(n_layers):
        hn = F.layer_norm(h, (d,), sd[f"norms.{i}.weight"], sd[f"norms.{i}.bias"])
        h_res = gcn_conv(hn, edge_index, sd[f"gcn_layers.{i}.lin.weight"], sd[f"gcn_layers.{i}.bias"])
        h = h + F.gelu(h_res * gate)
        states.append(h)
    s = sage_conv(h, edge_index, sd["sage.lin_l.weight"], sd["sage.lin_l.bias"], sd["sage.lin_r.weight"])
    states.append(F.gelu(F.layer_norm(s, (d,), sd["sage_norm.weight"], sd["sage_norm.bias"])))
    w = torch.softmax(sd["jk_logits"], dim=0)
    h_jk = torch.stack(states, 0).mul(w[:, None, None]).sum(0)
    a = graph_softmax(F.linear(h_jk, sd["ctx.attn.weight"], sd["ctx.attn.bias"]), batch)
    if batch is None:
        g = (a * h_jk).sum(0, keepdim=True)
    else:
        ng = int(batch.max()) + 1
        g = torch.zeros(ng, d).index_add_(0, batch, a * h_jk)[batch]
    g = torch.sigmoid(F.linear(F.relu(F.linear(g, sd["ctx.compress.weight"], sd["ctx.compress.bias"])),
                               sd["ctx.expand.weight"], sd["ctx.expand.bias"]))
    z = h_jk * g
    f = F.gelu(F.linear(F.layer_norm(z, (d,), sd["fuse.0.weight"], sd["fuse.0.bias"]),
                        sd["fuse.1.weight"], sd["fuse.1.bias"]))
    return F.linear(f, sd["head.weight"], sd["head.bias"])


@torch.no_grad()
def gcnnet_forward(sd, n_layers, x, edge_index, edge_attr):
    """GCNTrimapNet.forward in eval mode (reference model.py:239-316, blocks :216-232, edge injection :142-162).
    sd: state_dict of float32 CPU tensors with the reference keys."""
    def bn(v, p):
        return F.batch_norm(v, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)
    n = x.size(0)
    h = F.relu(bn(F.linear(bn(x, "in_norm.norm"), sd["input_proj.0.weight"], sd["input_proj.0.bias"]), "input_proj.1"))
    all_h = [h]
    for i in range(n_layers):
        p = f"blocks.{i}."
        c = F.relu(bn(gcn_conv(h, edge_index, sd[p + "conv.lin.weight"], sd[p + "conv.bias"]), p + "bn")) + h
        g = torch.sigmoid(F.linear(F.relu(F.linear(edge_attr, sd[p + "edge_inject.proj.0.weight"], sd[p + "edge_inject.proj.0.bias"])),
                                   sd[p + "edge_inject.proj.2.weight"], sd[p + "edge_inject.proj.2.bias"]))
        h = c * scatter_mean(g, edge_index[1], n)
        all_h.append(h)
    z = F.relu(bn(F.linear(torch.cat(all_h, dim=-1), sd["head.0.weight"], sd["head.0.bias"]), "head.1"))
    z = F.relu(F.linear(z, sd["head.4.weight"], sd["head.4.bias"]))
    logits = F.linear(z, sd["head.6.weight"], sd["head.6.bias"])
    return logits, torch.softmax(logits, dim=-1)


def gatv2_conv(x, edge_index, edge_attr, sd, p, heads):
    """torch_geometric.nn.GATv2Conv(D, D // heads, heads, concat=True, edge_dim=5, share_weights=False) in eval mode, from its
    documented semantics (add_self_loops=True with fill_value="mean", negative_slope 0.2, softmax over the edges into a node)."""
    n, d = x.size(0), sd[p + "lin_l.weight"].size(0)
    c = d // heads
    x_l = F.linear(x, sd[p + "lin_l.weight"], sd[p + "lin_l.bias"]).view(n, heads, c)
    x_r = F.linear(x, sd[p + "lin_r.weight"], sd[p + "lin_r.bias"]).view(n, heads, c)
    loops = torch.arange(n, dtype=edge_index.dtype)
    loop_attr = scatter_mean(edge_attr, edge_index[1], n)                  # self-loop attribute: mean of the incoming ones
    src = torch.cat([edge_index[0], loops]); dst = torch.cat([edge_index[1], loops])
    ea = torch.cat([edge_attr, loop_attr], 0)
    m = x_r[dst] + x_l[src] + F.linear(ea, sd[p + "lin_edge.weight"]).view(-1, heads, c)
    m = F.leaky_relu(m, 0.2)
    alpha = (m * sd[p + "att"]).sum(-1)                                    # (E + N, heads)
    peak = torch.full((n, heads), float("-inf")).index_reduce(0, dst, alpha, "amax", include_self=True)
    ex = torch.exp(alpha - peak[dst])
    tot = torch.zeros(n, heads).index_add_(0, dst, ex)
    alpha = ex / (tot[dst] + 1e-16)
    out = torch.zeros(n, heads, c).index_add_(0, dst, alpha.unsqueeze(-1) * x_l[src])
    return out.reshape(n, d) + sd[p + "bias"]


@torch.no_grad()
def gat_forward(sd, n_layers, x, edge_index, edge_attr, batch=None, heads=8):
    """GATTrimapNet.forward in eval mode (reference model.py:380-404).  sd: state_dict of float32 CPU tensors, reference keys."""
    n = x.size(0)
    d = sd["input_proj.0.weight"].size(0)
    xn = F.batch_norm(x, sd["in_norm.norm.running_mean"], sd["in_norm.norm.running_var"],
                      sd["in_norm.norm.weight"], sd["in_norm.norm.bias"], False, 0.0, 1e-5)
    h = F.gelu(F.layer_norm(F.linear(xn, sd["input_proj.0.weight"], sd["input_proj.0.bias"]), (d,),
                            sd["input_proj.1.weight"], sd["input_proj.1.bias"]))
    skip = F.linear(h, sd["skip_proj.weight"])
    for i in range(n_layers):
        hn = F.gelu(F.layer_norm(gatv2_conv(h, edge_index, edge_attr, sd, f"convs.{i}.", heads), (d,), sd[f"lns.{i}.weight"], sd[f"lns.{i}.bias"]))
        g = torch.sigmoid(F.linear(F.relu(F.linear(edge_attr, sd[f"edge_gates.{i}.proj.0.weight"], sd[f"edge_gates.{i}.proj.0.bias"])),
                                   sd[f"edge_gates.{i}.proj.2.weight"], sd[f"edge_gates.{i}.proj.2.bias"]))
        h = hn * scatter_mean(g, edge_index[1], n)
    h = h + skip
    a = graph_softmax(F.linear(h, sd["ctx.attn.weight"], sd["ctx.attn.bias"]), batch)
    if batch is None:
        g = (a * h).sum(0, keepdim=True)
    else:
        ng = int(batch.max()) + 1
        g = torch.zeros(ng, d).index_add_(0, batch, a * h)[batch]
    g = torch.sigmoid(F.linear(F.relu(F.linear(g, sd["ctx.compress.weight"], sd["ctx.compress.bias"])),
                               sd["ctx.expand.weight"], sd["ctx.expand.bias"]))
    z = F.gelu(F.linear(h * g, sd["head.0.weight"], sd["head.0.bias"]))
    logits = F.linear(z, sd["head.3.weight"], sd["head.3.bias"])
    return logits, torch.softmax(logits, dim=-1)
