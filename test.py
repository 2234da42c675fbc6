#!/usr/bin/env python
"""Inference entry point on the MI355X hot path (counterpart of the reference's test.py:10-33).

    python test.py --yaml=test --name=run --nerf.rand_rays_test=4096 --nerf.sample_intvs=64

Options use the reference's ``--a.b.c=value`` grammar and YAML inheritance; the configured
test sets are served by seeded synthetic scenes when no dataset is on disk."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def run(argv):
    from matchnerf_amd import options
    from matchnerf_amd.coach import Coach

    opt = options.set(opt_cmd=options.parse_arguments(argv))
    options.save_options_file(opt)
    coach = Coach(opt)
    coach.build_networks()
    coach.restore_checkpoint()
    coach.load_dataset(splits=["test"])
    if opt.nerf.render_video:
        return coach.test_model_video()
    return coach.test_model(save_images=bool(getattr(opt, "separate_save", False)))


if __name__ == "__main__":
    run(sys.argv[1:])
