"""Shared helpers for the parity tests (oracle side)."""
import numpy as np
import torch

from conftest import load_golden  # noqa: F401
from matchnerf_amd import synthetic as syn
from oracle import matchnerf_oracle as O


def effective_overrides(meta):
    """The golden's dotted option overrides on top of what its yaml (meta['yaml'], default test.yaml) changes relative to
    test.yaml — read from this repository's own option trees (pinned equal to the reference's by tests/golden/options.json)."""
    ov = dict(meta["opt_overrides"])
    name = meta.get("yaml", "test")
    if name != "test":
        from matchnerf_amd import options
        opt = options.load_options(f"configs/{name}.yaml", verbose=False)
        base = {"nerf.sample_intvs": opt.nerf.sample_intvs, "n_src_views": opt.n_src_views,
                "decoder.density_maskfill": opt.decoder.density_maskfill, "decoder.raytrans_posenc": opt.decoder.raytrans_posenc,
                "decoder.raytrans_act": opt.decoder.raytrans_act, "nerf.legacy_coord": opt.nerf.legacy_coord,
                "nerf.wo_render_interval": opt.nerf.wo_render_interval, "nerf.depth.param": opt.nerf.depth.param,
                "encoder.attn_splits_list": list(opt.encoder.attn_splits_list)}
        base.update(ov)
        ov = base
    return ov


def cfg_from_meta(meta):
    ov = effective_overrides(meta)
    return O.OracleConfig(
        sample_intvs=ov.get("nerf.sample_intvs", 128), n_src_views=ov.get("n_src_views", 3),
        density_maskfill=ov.get("decoder.density_maskfill", False),
        raytrans_posenc=ov.get("decoder.raytrans_posenc", False),
        raytrans_act=ov.get("decoder.raytrans_act", "ReLU"),
        legacy_coord=ov.get("nerf.legacy_coord", True),
        wo_render_interval=ov.get("nerf.wo_render_interval", True),
        depth_param=ov.get("nerf.depth.param", "metric"),
        attn_splits=ov.get("encoder.attn_splits_list", [2])[0])


def golden_case(name):
    """-> (golden dict, OracleConfig, torch state_dict, torch batch)"""
    g = load_golden(name)
    cfg = cfg_from_meta(g["meta"])
    w = syn.seeded_state_dict(syn.state_dict_spec(n_src_views=cfg.n_src_views), g["meta"]["weight_seed"])
    sd = syn.to_torch(w)
    batch = {k: torch.from_numpy(g[k]) for k in ("images", "extrinsics", "intrinsics", "near_fars")}
    return g, cfg, sd, batch


def split_poses(batch, b=0):
    te, ti, tn = batch["extrinsics"][b, -1, :3], batch["intrinsics"][b, -1], batch["near_fars"][b, -1]
    se, si, sn = batch["extrinsics"][b, :-1, :3], batch["intrinsics"][b, :-1], batch["near_fars"][b, :-1]
    return te, ti, tn, se, si, sn


def linf(a, b):
    a = torch.as_tensor(np.asarray(a)) if not torch.is_tensor(a) else a
    b = torch.as_tensor(np.asarray(b)) if not torch.is_tensor(b) else b
    return float((a.float().cpu() - b.float().cpu()).abs().max())
