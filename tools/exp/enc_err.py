"""Where does the encoder's deviation from the reference come from?  Per golden case: L-inf of the backbone output, of the two
feature scales (GPU vs the reference's goldens), for the attention matrix paths.  usage: enc_err.py [case ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch

from helpers import golden_case, linf
from test_model_gpu import build_model, to_batch
from matchnerf_amd.gmflow import pair_major_to_view_chunks

for name in sys.argv[1:] or ["c1_default", "rect_wide", "demo_own_small", "demo_own"]:
    g, cfg, sd, _ = golden_case(name)
    opt, model = build_model(g["meta"])
    batch = to_batch(g)
    cap = {}
    h = model.feat_enc.backbone.register_forward_hook(lambda m, i, o: cap.setdefault("bb", (o[0] if isinstance(o, (list, tuple)) else o).detach()))
    with torch.no_grad():
        feats = model.get_img_feat(batch.images[:, :cfg.n_src_views], cur_n_src_views=cfg.n_src_views)
    h.remove()
    line = f"{name} wa_math={os.environ.get('MNERF_WA_MATH', 'default')}:"
    if "bb" in cap:
        bb = cap["bb"]
        ref = torch.from_numpy(g["backbone"])
        if bb.shape != ref.shape and bb.dim() == 4 and bb.shape[-1] == ref.shape[1]:
            bb = bb.permute(0, 3, 1, 2)
        if bb.shape == ref.shape:
            line += f" backbone {linf(bb, ref):.2e} (max {float(ref.abs().max()):.1f})"
        else:
            line += f" backbone shape {tuple(bb.shape)} vs {tuple(ref.shape)}"
    for i, f in enumerate(feats):
        rl = pair_major_to_view_chunks(f)[0].cpu()
        a, b = (rl, g[f"feat_scale{i}"]) if f"feat_scale{i}" in g else (rl[:, ::16], g[f"feat_scale{i}_sub"])
        line += f" scale{i} {linf(a, b):.2e} (max {float(np.abs(b).max()):.1f})"
    print(line, flush=True)
