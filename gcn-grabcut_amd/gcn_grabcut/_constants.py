"""Feature widths shared by graph_builder and model (reference graph_builder.py:73-77)."""
N_IMAGE_FEATS = 16   # image-derived node features
N_PRIOR_FEATS = 3    # automatic FG / BG / ambiguity prior
N_HINT_FEATS = N_PRIOR_FEATS   # backwards-compatible alias
N_NODE_FEATS = N_IMAGE_FEATS + N_PRIOR_FEATS
N_EDGE_FEATS = 5
