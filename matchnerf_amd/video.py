"""Camera trajectories for video rendering — host-side numpy/scipy, as in the reference
(/root/reference/misc/camera.py:382-468).  They only produce the list of target poses that
feeds the same HIP render loop; pinned by tests/golden/video_paths.npz."""
import numpy as np
from scipy.spatial.transform import Rotation


def interpolate_render_path(c2ws, n_views=30):
    """Closed loop through the source cameras: xyz-Euler angles (degrees, unwrapped against the
    first camera) and positions are blended linearly, n_views//3 steps per leg
    (camera.py:382-411)."""
    n = len(c2ws)
    w = np.linspace(1.0, 0.0, n_views // 3, endpoint=False).reshape(-1, 1)
    angles, positions = [], []
    for i in range(n):
        e = Rotation.from_matrix(c2ws[i, :3, :3]).as_euler("xyz", degrees=True).reshape(1, 3)
        if i:
            e[np.abs(e - angles[0]) > 180] += 360.0
        angles.append(e)
        positions.append(c2ws[i, :3, 3].reshape(1, 3))
    legs = [(i - 1, i) for i in range(1, n)] + [(n - 1, 0)]
    ang = np.concatenate([w * angles[a] + (1.0 - w) * angles[b] for a, b in legs])
    pos = np.concatenate([w * positions[a] + (1.0 - w) * positions[b] for a, b in legs])
    out = []
    for e, p in zip(ang, pos):
        m = np.eye(4)
        m[:3, :3] = Rotation.from_euler("xyz", e, degrees=True).as_matrix()
        m[:3, 3] = p
        out.append(m)
    return np.stack(out)


def _unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def _view_matrix(z, up, pos):
    z = _unit(z)
    x = _unit(np.cross(up, z))
    y = _unit(np.cross(z, x))
    m = np.eye(4)
    m[:3] = np.stack([x, y, z, pos], 1)
    return m


def spiral_render_path(c2ws_all, near_far, rads_scale=0.5, n_views=120, n_rots=2, zrate=0.5):
    """LLFF-style spiral around the average pose (camera.py:415-468)."""
    center = c2ws_all[:, :3, 3].mean(0)
    c2w = _view_matrix(_unit(c2ws_all[:, :3, 2].sum(0)), c2ws_all[:, :3, 1].sum(0), center)
    up = _unit(c2ws_all[:, :3, 1].sum(0))
    close_depth, inf_depth = near_far
    dt = 0.75
    focal = 1.0 / ((1.0 - dt) / close_depth + dt / inf_depth)
    rads = np.percentile(np.abs(c2ws_all[:, :3, 3] - c2w[:3, 3][None]), 70, 0) * rads_scale
    rads = np.array(list(rads) + [1.0])
    out = []
    for theta in np.linspace(0.0, 2.0 * np.pi * n_rots, n_views + 1)[:-1]:
        c = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.0]) * rads)
        z = _unit(c - np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.0])))
        out.append(_view_matrix(z, up, c))
    return np.stack(out)
