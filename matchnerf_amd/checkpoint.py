"""Checkpoint conventions of the reference (misc/utils.py:156-222, coach.py:290-300):
``{model, optim, sched, epoch, iter}`` in ``<dir>/models/latest.pth`` (+ ``ep{E}_it{I}.pth``
without optimizer state); restore is per top-level child, strict — so a reference checkpoint
(``matchnerf_3v.pth``) loads into this module unchanged."""
import os

import torch


def child_state_dict(state_dict, key):
    prefix = key + "."
    return {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}


def restore_checkpoint(model, ckpt_path, device, resume=False, optims_scheds=None, log=print):
    ckpt = torch.load(ckpt_path, map_location=device)
    for name, child in model.named_children():
        sd = child_state_dict(ckpt["model"], name)
        if sd:
            child.load_state_dict(sd, strict=True)
            log(f"  * restored {name} from {ckpt_path}")
    if not resume:
        return None, None
    assert optims_scheds is not None, "Must provide full optims (and / or scheds) for resume training."
    for name, obj in optims_scheds.items():
        if name in ckpt:
            obj.load_state_dict(ckpt[name])
    return ckpt["epoch"], ckpt["iter"]


def load_gmflow_checkpoint(model_enc, ckpt_path, device, gmflow_n_blocks=6):
    """Pre-trained GMFlow weights into backbone + transformer (misc/utils.py:160-180): drops
    GMFlow's own up-sampler / refinement attention and transformer layers beyond n_blocks."""
    ckpt = torch.load(ckpt_path, map_location=device)
    weights = ckpt["model"] if "model" in ckpt else ckpt
    keep = {}
    for k, v in weights.items():
        if k.startswith("upsampler") or k.startswith("feature_flow_attn"):
            continue
        if any(k.startswith(f"transformer.layers.{i}.") for i in range(gmflow_n_blocks, 6)):
            continue
        keep[k] = v
    for name, child in model_enc.named_children():
        if name != "featup_net":
            child.load_state_dict(child_state_dict(keep, name), strict=True)


def save_checkpoint(saved_dir, checkpoint, ep, it, backup_ckpt=True, children=None):
    """misc/utils.py:208-222: ``epoch`` / ``iter`` are stamped into the file here (callers pass only
    model / optim / sched), ``children`` keeps the model keys with that prefix (str or tuple)."""
    ckpt_dir = os.path.join(saved_dir, "models")
    os.makedirs(ckpt_dir, exist_ok=True)
    model_sd = checkpoint["model"]
    if children is not None:
        model_sd = {k: v for k, v in model_sd.items() if k.startswith(children)}
    full = dict(checkpoint, epoch=ep, iter=it, model=model_sd)
    latest = os.path.join(ckpt_dir, "latest.pth")
    torch.save(full, latest)
    if backup_ckpt:
        slim = {k: v for k, v in full.items() if k not in ("optim", "sched")}
        torch.save(slim, os.path.join(ckpt_dir, f"ep{ep}_it{it}.pth"))
    return latest
