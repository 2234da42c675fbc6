#!/usr/bin/env python
"""Counterpart of the reference's test.py (test.py:10-33):
    python test.py --yaml=test --name=run --nerf.rand_rays_test=4096 --nerf.sample_intvs=64
Runs the MI355X hot path over the configured test sets (synthetic stand-ins offline)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from matchnerf_amd import options  # noqa: E402
from matchnerf_amd.coach import Coach  # noqa: E402


def main():
    opt_cmd = options.parse_arguments(sys.argv[1:])
    opt = options.set(opt_cmd=opt_cmd)
    options.save_options_file(opt)
    m = Coach(opt)
    m.build_networks()
    m.restore_checkpoint()
    m.load_dataset(splits=["test"])
    if opt.nerf.render_video:
        m.test_model_video()
    else:
        m.test_model(save_images=bool(getattr(opt, "separate_save", False)))


if __name__ == "__main__":
    main()
