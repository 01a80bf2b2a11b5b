"""Render-trajectory generators (host side, numpy) for `Runner.render_traj`.

Behavioural restatement of the reference's utils/datasets/traj.py (itself after multinerf):
  viewmatrix :16-22, focus_point_fn :25-31, average_pose :34-40, generate_spiral_path :43-79,
  generate_ellipse_path_z :82-141, generate_ellipse_path_y :144-203,
  generate_interpolated_path :206-255.
All take / return [n,3,4] (or [n,4,4] on input) camera-to-world matrices.
Pinned by tests/golden/g4_traj.npz.
"""
import numpy as np


def _unit(v):
    return v / np.linalg.norm(v)


def viewmatrix(lookdir, up, position):
    """[3,4] pose whose z column is `lookdir`, y column the part of `up` orthogonal to it."""
    z = _unit(np.asarray(lookdir, dtype=np.float64))
    x = _unit(np.cross(up, z))
    y = _unit(np.cross(z, x))
    return np.stack([x, y, z, np.asarray(position, dtype=np.float64)], axis=1)


def focus_point_fn(poses):
    """Least-squares point nearest to every optical axis."""
    d = poses[:, :3, 2]
    o = poses[:, :3, 3]
    P = np.eye(3)[None] - d[:, :, None] * d[:, None, :]     # projector orthogonal to each axis
    A = np.einsum("nji,njk->nik", P, P)
    return np.linalg.inv(A.mean(0)) @ np.einsum("nij,nj->ni", A, o).mean(0)


def average_pose(poses):
    return viewmatrix(poses[:, :3, 2].mean(0), poses[:, :3, 1].mean(0), poses[:, :3, 3].mean(0))


def generate_spiral_path(poses, bounds, n_frames=120, n_rots=2, zrate=0.5, spiral_scale_f=1.0,
                         spiral_scale_r=1.0, focus_distance=0.75):
    """Forward-facing spiral about the average pose; cameras look at a point `focal` ahead."""
    near, far = bounds.min(), bounds.max()
    focal = spiral_scale_f / ((1.0 - focus_distance) / near + focus_distance / far)
    radii = np.append(np.percentile(np.abs(poses[:, :3, 3]), 90, 0) * spiral_scale_r, 1.0)
    centre = average_pose(poses)
    up = poses[:, :3, 1].mean(0)
    target = centre @ np.array([0.0, 0.0, -focal, 1.0])
    out = []
    for th in np.linspace(0.0, 2.0 * np.pi * n_rots, n_frames, endpoint=False):
        p = centre @ (radii * np.array([np.cos(th), -np.sin(th), -np.sin(th * zrate), 1.0]))
        out.append(viewmatrix(p - target, up, p))
    return np.stack(out, axis=0)


def _ellipse_path(poses, n_frames, variation, phase, height, plane, sign):
    """Shared body of the _z / _y variants: ellipse in the two axes of `plane`, third axis `h`."""
    a, b = plane
    h = 3 - a - b
    centre = focus_point_fn(poses)
    offset = centre.copy()
    offset[h] = height
    cam = poses[:, :3, 3]
    half = np.percentile(np.abs(cam - offset), 90, axis=0)
    lo, hi = offset - half, offset + half
    h_lo, h_hi = np.percentile(cam, 10, axis=0)[h], np.percentile(cam, 90, axis=0)[h]

    th = np.linspace(0.0, 2.0 * np.pi, n_frames + 1, endpoint=True)[:-1]
    pos = np.empty((n_frames, 3))
    pos[:, a] = lo[a] + (hi - lo)[a] * (np.cos(th) * 0.5 + 0.5)
    pos[:, b] = lo[b] + (hi - lo)[b] * (np.sin(th) * 0.5 + 0.5)
    pos[:, h] = variation * (h_lo + (h_hi - h_lo) * (np.cos(th + 2.0 * np.pi * phase) * 0.5 + 0.5)) + height

    mean_up = _unit(poses[:, :3, 1].mean(0))
    k = int(np.argmax(np.abs(mean_up)))
    up = np.eye(3)[k] * np.sign(mean_up[k])
    return np.stack([viewmatrix(sign * (centre - p), up, p) for p in pos])


def generate_ellipse_path_z(poses, n_frames=120, variation=0.0, phase=0.0, height=0.0):
    """Ellipse in x-y at z=height; the z column points from the camera TO the focus (traj.py:141)."""
    return _ellipse_path(poses, n_frames, variation, phase, height, (0, 1), +1.0)


def generate_ellipse_path_y(poses, n_frames=120, variation=0.0, phase=0.0, height=0.0):
    """Ellipse in x-z at y=height; the z column points AWAY from the focus (traj.py:203)."""
    return _ellipse_path(poses, n_frames, variation, phase, height, (0, 2), -1.0)


def generate_interpolated_path(poses, n_interp, spline_degree=5, smoothness=0.03, rot_weight=0.1):
    """Smoothing B-spline through (position, look-at point, up point) triples of the key frames;
    returns n_interp*(n-1) poses."""
    import scipy.interpolate

    p = poses[:, :3, 3]
    pts = np.stack([p, p - rot_weight * poses[:, :3, 2], p + rot_weight * poses[:, :3, 1]], axis=1)
    n = pts.shape[0]
    m = n_interp * (n - 1)
    tck, _ = scipy.interpolate.splprep(pts.reshape(n, -1).T, k=min(spline_degree, n - 1), s=smoothness)
    new = np.array(scipy.interpolate.splev(np.linspace(0.0, 1.0, m, endpoint=False), tck)).T.reshape(m, 3, 3)
    return np.array([viewmatrix(q[0] - q[1], q[2] - q[0], q[0]) for q in new])
