"""Drop-in check against the reference's OWN coach.py — BUILD CONTAINER ONLY (needs /root/reference).

    python tools/ref_coach_dropin.py

Imports the reference's ``coach.py`` in place (``tools/ref_import.py`` stubs for absent non-path modules),
swaps ``models_dict['matchnerf']`` for this repo's MatchNeRF — the one-line change INTEGRATION.md describes —
and drives the reference's unmodified ``Coach.build_networks`` / ``setup_optimizer`` / ``restore_checkpoint`` /
``save_checkpoint`` on the CPU with ``gpu_ids=[0]`` and ``gpu_ids=[0, 1]`` (the nn.DataParallel wrapping of
coach.py:83-85).  Checks: the reference's strict per-child checkpoint restore accepts the module, the optimizer
sees the same parameter groups, the module still finds its decoder through the DataParallel wrapper, and a
checkpoint written by the reference's save path restores bit-exactly.  (Rendering needs the GPU: tests -m gpu.)
"""
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from ref_import import _stub, import_reference, reference_options  # noqa: E402
from matchnerf_amd import synthetic as syn  # noqa: E402
from matchnerf_amd.matchnerf import MatchNeRF  # noqa: E402


def main():
    import_reference()
    _stub("imageio")
    _stub("lpips", LPIPS=lambda *a, **k: None)
    sk = _stub("skimage")
    sk.metrics = _stub("skimage.metrics", structural_similarity=lambda *a, **k: 0.0)
    import coach as ref_coach  # the reference's coach.py, unmodified
    ref_coach.models_dict["matchnerf"] = MatchNeRF  # the one-line registry change
    weights = syn.to_torch(syn.seeded_state_dict(syn.state_dict_spec(), 3))
    for gpu_ids in ([0], [0, 1]):
        with tempfile.TemporaryDirectory() as tmp:
            opt = reference_options("train", gpu_ids=gpu_ids, output_path=tmp, resume=False, load=None)
            opt.optim.sched = None                      # OneCycleLR needs a train loader (datasets: out of scope)
            opt.encoder.pretrain_weight = None
            c = ref_coach.Coach(opt)
            c.build_networks()
            assert isinstance(c.model, MatchNeRF)
            wrapped = len(gpu_ids) > 1
            assert isinstance(c.model.nerf_dec, torch.nn.DataParallel) == wrapped
            assert c.model._dec().cond_dim == 22 and c.model._decoder(64, torch.device("cpu")).cond_stride == 24
            c.setup_optimizer()
            n_opt = sum(p.numel() for g in c.optim.param_groups for p in g["params"])
            assert n_opt == sum(p.numel() for p in c.model.parameters()), "optimizer must see every parameter"
            assert [g["lr"] for g in c.optim.param_groups] == [opt.optim.lr_enc, opt.optim.lr_dec]
            # a checkpoint in the reference's format (what matchnerf_3v.pth looks like), keys as THIS run would save them
            prefix = "module." if wrapped else ""
            sd = {k.replace("feat_enc.", "feat_enc." + prefix, 1).replace("nerf_dec.", "nerf_dec." + prefix, 1): v
                  for k, v in weights.items()}
            ck = os.path.join(tmp, "w.pth")
            torch.save(dict(model=sd), ck)
            opt.load = ck
            c.restore_checkpoint()                      # misc/utils.py:183-205: strict, per child
            got = c.model.state_dict()
            assert list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
            c.save_checkpoint(ep=2, it=11)              # coach.py:290-300 + misc/utils.py:208-222
            saved = torch.load(os.path.join(tmp, "models", "latest.pth"))
            assert saved["epoch"] == 2 and saved["iter"] == 11 and "optim" in saved
            assert all(torch.equal(saved["model"][k], sd[k]) for k in sd)
            # ... and this repo's restore reads the file the reference wrote
            from matchnerf_amd import checkpoint
            m2 = MatchNeRF(opt)
            if wrapped:
                m2.feat_enc = torch.nn.DataParallel(m2.feat_enc, gpu_ids)
                m2.nerf_dec = torch.nn.DataParallel(m2.nerf_dec, gpu_ids)
            checkpoint.restore_checkpoint(m2, os.path.join(tmp, "models", "ep2_it11.pth"), "cpu", log=lambda *a: None)
            assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), sd.values()))
            print(f"[ref-coach] gpu_ids={gpu_ids}: build_networks / setup_optimizer / restore_checkpoint / save_checkpoint OK "
                  f"({len(sd)} tensors, {n_opt} optimised parameters)")


if __name__ == "__main__":
    main()
