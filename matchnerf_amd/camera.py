"""Host-side camera algebra of the hot path (tiny 3x3 / 4x4 work, done once per target pose).

The per-ray arithmetic of the reference's misc/camera.py (get_center_and_ray 255-278,
get_3D_points_from_depth 281-286, get_coord_ref_ndc 351-379) lives inside the HIP kernels
(csrc/common.hpp); what stays on the host is what the reference also computes once per call:
the inverse intrinsics and the camera->world matrix of the target view.
"""
import numpy as np
import torch


def target_ray_consts(extr34, intr33, legacy=True):
    """-> (kinv [3,3], c2w [3,4]) float32 numpy.

    kinv: fp32 ``intr.inverse()`` (camera.py:221-222).
    c2w : legacy — inverse of the 4x4 world->cam taken in float64 then cast to float32
          (cam2world_legacy, camera.py:231-240); otherwise [R^T | -R^T t] in float32
          (Pose.invert, camera.py:36-42)."""
    extr = torch.as_tensor(extr34, dtype=torch.float32).detach().cpu().reshape(3, 4)
    intr = torch.as_tensor(intr33, dtype=torch.float32).detach().cpu().reshape(3, 3)
    kinv = intr.inverse()
    if legacy:
        sq = torch.eye(4)
        sq[:3] = extr
        c2w = sq.double().inverse()[:3].float()
    else:
        rot_inv = extr[:, :3].t()
        c2w = torch.cat([rot_inv, -(rot_inv @ extr[:, 3:])], 1)
    return kinv.numpy().copy(), c2w.contiguous().numpy().copy()


def pair_list(n_views):
    """Ordered view pairs (a<b) (gmflow.py:49, matchnerf.py:194)."""
    return [(a, b) for a in range(n_views - 1) for b in range(a + 1, n_views)]
