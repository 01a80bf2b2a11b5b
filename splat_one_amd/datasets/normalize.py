"""World-space normalisation of an SfM reconstruction (host side, float64 numpy).

Restates the behaviour of the reference's utils/datasets/normalize.py:
  similarity_from_cameras  :4-63    up-axis alignment, "focus"/"poses" recentring, median/max rescale
  align_principle_axes     :66-101  PCA of the point cloud about its per-axis median
  transform_points         :104-116
  transform_cameras        :119-134 (re-normalises the rotation block by the norm of its first row)
  normalize                :137-146
Pinned by tests/golden/g3_normalize.npz (outputs of the reference functions themselves).
"""
import numpy as np

_CAM_UP = np.array([0.0, -1.0, 0.0])  # OpenCV cameras: -y is up


def _rotation_between(a, b):
    """Smallest rotation taking unit vector `a` onto unit vector `b` (Rodrigues, 1/(1+cos) form)."""
    cos = float(a @ b)
    if cos <= -1.0:
        # antiparallel: the reference picks the half-turn that negates x (normalize.py:33-36)
        return np.diag([-1.0, 1.0, 1.0])
    ax = np.cross(a, b)
    S = np.array([[0.0, -ax[2], ax[1]], [ax[2], 0.0, -ax[0]], [-ax[1], ax[0], 0.0]])
    return np.eye(3) + S + S @ S / (1.0 + cos)


def similarity_from_cameras(c2w, strict_scaling=False, center_method="focus"):
    """4x4 similarity (rows 0..2 scaled) that puts +z up, the scene focus at the origin and the
    median (or max, `strict_scaling`) camera distance at 1.  c2w: [N,4,4] OpenCV convention."""
    c2w = np.asarray(c2w, dtype=np.float64)
    rot, pos = c2w[:, :3, :3], c2w[:, :3, 3]

    up = (rot @ _CAM_UP).mean(axis=0)
    up /= np.linalg.norm(up)
    R_align = _rotation_between(up, _CAM_UP)

    fwd = (R_align @ rot)[:, :, 2]            # camera +z in the aligned frame
    pos = pos @ R_align.T
    if center_method == "focus":
        # point of each optical axis closest to the origin
        foot = pos - np.sum(fwd * pos, axis=-1, keepdims=True) * fwd
        shift = -np.median(foot, axis=0)
    elif center_method == "poses":
        shift = -np.median(pos, axis=0)
    else:
        raise ValueError(f"Unknown center_method {center_method}")

    dist = np.linalg.norm(pos + shift, axis=-1)
    s = 1.0 / (dist.max() if strict_scaling else np.median(dist))
    T = np.eye(4)
    T[:3, :3] = s * R_align
    T[:3, 3] = s * shift
    return T


def align_principle_axes(point_cloud):
    """SE(3) taking the cloud's median to the origin and its principal axes to x,y,z by decreasing
    variance (so the thinnest direction becomes z)."""
    pts = np.asarray(point_cloud, dtype=np.float64)
    mid = np.median(pts, axis=0)
    evals, evecs = np.linalg.eigh(np.cov(pts - mid, rowvar=False))
    evecs = evecs[:, np.argsort(evals)[::-1]]
    if np.linalg.det(evecs) < 0:
        evecs[:, 0] = -evecs[:, 0]
    T = np.eye(4)
    T[:3, :3] = evecs.T
    T[:3, 3] = -evecs.T @ mid
    return T


def transform_points(matrix, points):
    assert matrix.shape == (4, 4)
    assert points.ndim == 2 and points.shape[1] == 3
    return points @ matrix[:3, :3].T + matrix[:3, 3]


def transform_cameras(matrix, camtoworlds):
    """Left-multiply every c2w by `matrix`, then strip the similarity's scale from the rotation
    block (the translation keeps it)."""
    assert matrix.shape == (4, 4)
    assert camtoworlds.ndim == 3 and camtoworlds.shape[1:] == (4, 4)
    out = matrix[None] @ camtoworlds
    scale = np.linalg.norm(out[:, 0, :3], axis=1)
    out[:, :3, :3] /= scale[:, None, None]
    return out


def normalize(camtoworlds, points=None):
    """normalize.py:137-146.  Returns (camtoworlds, points, T) or (camtoworlds, T1) without points."""
    T1 = similarity_from_cameras(camtoworlds)
    camtoworlds = transform_cameras(T1, camtoworlds)
    if points is None:
        return camtoworlds, T1
    points = transform_points(T1, points)
    T2 = align_principle_axes(points)
    return transform_cameras(T2, camtoworlds), transform_points(T2, points), T2 @ T1
