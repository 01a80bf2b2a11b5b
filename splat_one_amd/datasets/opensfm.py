"""OpenSfM `reconstruction.json` ingestion -- the data format on the input side of the hot path
(SURVEY.md section 8 row f3).  Host side, numpy; nothing here touches the GPU.

Behaviour follows the reference's utils/datasets/opensfm.py:
  angle_axis_to_quaternion :72-84, qvec2rotmat :47-57            shot rotation -> w2c rotation
  read_opensfm            :400-464   cameras / shots of every reconstruction in the file
  read_opensfm_points3D   :466-501   point cloud, shifted into the first reconstruction's frame
  Parser                  :121-311   K, c2w = inv(w2c), optional normalisation, undistortion, scene_scale
  Dataset                 :313-397   train/val split by index % test_every, item dict
The reference needs pyproj, cv2 and imageio for this; none is in the image, so the three
things they provide are restated here from their published definitions:
  * `utm_forward`   WGS84 transverse Mercator (Krueger n-series, 6th order) for the reference-LLA
                    offsets between reconstructions (zero for a single reconstruction);
  * `undistort_maps` OpenCV's getOptimalNewCameraMatrix(alpha=0) + initUndistortRectifyMap for the
                    (k1,k2,0,0) radial model, `remap_bilinear` for cv2.remap(INTER_LINEAR);
                    UNPINNED (no cv2 here to compare with): float arithmetic, so 8-bit results may
                    differ from cv2's 5-bit fixed-point interpolation by one level;
  * `resize_area`   cv2.resize(INTER_AREA) as exact fractional box coverage.

Deliberate differences from the reference ("know the quirks, do not replicate
blindly"):
  * shots are keyed by a running index over all reconstructions; the reference keys them by the
    index inside their reconstruction (:451), so a second reconstruction overwrites the first;
  * a zero rotation vector gives the identity quaternion instead of NaN (:75-77 divide by |r|=0);
  * southern-hemisphere references do not raise (the reference reads `reference_y` before assigning
    it at :486-487); the false northing cancels in the differences either way;
  * the undistortion branch is chosen per camera, not by the model of the last parsed shot (:239);
  * unsupported projection types raise instead of silently mapping to camera id 0.
The spherical camera keeps the reference's K layout (:188) for callers that inspect it; the rasteriser's
`spherical` model (csrc/splat_math.hpp, defined by this build) takes width and height only and ignores that K.
"""
from __future__ import annotations

import collections
import json
import math
import os
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .normalize import align_principle_axes, similarity_from_cameras, transform_cameras, transform_points

Camera = collections.namedtuple("Camera", ["id", "model", "width", "height", "params", "panorama"])
_ImageBase = collections.namedtuple(
    "Image", ["id", "qvec", "tvec", "camera_id", "name", "xys", "point3D_ids", "diff_ref"])


class Image(_ImageBase):
    def qvec2rotmat(self):
        return qvec2rotmat(self.qvec)


# ----------------------------------------------------------------------------- rotations
def angle_axis_to_quaternion(angle_axis) -> np.ndarray:
    """Rotation vector -> (w,x,y,z).  The vector part is axis*sqrt(1-w^2), as the reference writes it."""
    r = np.asarray(angle_axis, dtype=np.float64)
    angle = float(np.linalg.norm(r))
    if angle == 0.0:
        return np.array([1.0, 0.0, 0.0, 0.0])
    w = math.cos(0.5 * angle)
    s = math.sqrt(1.0 - w * w)
    return np.array([w, r[0] / angle * s, r[1] / angle * s, r[2] / angle * s])


def qvec2rotmat(q) -> np.ndarray:
    w, x, y, z = (float(v) for v in q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (z * x + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (z * x - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rotmat2qvec(R) -> np.ndarray:
    """Largest-eigenvector method (opensfm.py:59-70); w >= 0."""
    R = np.asarray(R, dtype=np.float64)
    xx, yx, zx, xy, yy, zy, xz, yz, zz = R.reshape(-1)
    Kq = np.array([[xx - yy - zz, 0, 0, 0],
                   [yx + xy, yy - xx - zz, 0, 0],
                   [zx + xz, zy + yz, zz - xx - yy, 0],
                   [yz - zy, zx - xz, xy - yx, xx + yy + zz]]) / 3.0
    vals, vecs = np.linalg.eigh(Kq)
    q = vecs[[3, 0, 1, 2], int(np.argmax(vals))]
    return -q if q[0] < 0 else q


# ----------------------------------------------------------------------------- UTM
_WGS84_A = 6378137.0
_WGS84_F = 1.0 / 298.257223563


def utm_zone(lon: float) -> int:
    return int(lon // 6) + 31                                               # opensfm.py:407


def utm_forward(lon_deg: float, lat_deg: float, zone: int):
    """WGS84 UTM easting/northing (no false northing: pyproj's Proj(proj='utm') without `south`)."""
    n = _WGS84_F / (2.0 - _WGS84_F)
    A = _WGS84_A / (1.0 + n) * (1.0 + n ** 2 / 4.0 + n ** 4 / 64.0 + n ** 6 / 256.0)
    al = (n / 2 - 2 * n ** 2 / 3 + 5 * n ** 3 / 16 + 41 * n ** 4 / 180 - 127 * n ** 5 / 288 + 7891 * n ** 6 / 37800,
          13 * n ** 2 / 48 - 3 * n ** 3 / 5 + 557 * n ** 4 / 1440 + 281 * n ** 5 / 630 - 1983433 * n ** 6 / 1935360,
          61 * n ** 3 / 240 - 103 * n ** 4 / 140 + 15061 * n ** 5 / 26880 + 167603 * n ** 6 / 181440,
          49561 * n ** 4 / 161280 - 179 * n ** 5 / 168 + 6601661 * n ** 6 / 7257600,
          34729 * n ** 5 / 80640 - 3418889 * n ** 6 / 1995840,
          212378941 * n ** 6 / 319334400)
    lat = math.radians(lat_deg)
    dlon = math.radians(lon_deg - (zone * 6.0 - 183.0))
    e = math.sqrt(_WGS84_F * (2.0 - _WGS84_F))
    t = math.sinh(math.atanh(math.sin(lat)) - e * math.atanh(e * math.sin(lat)))   # tan of conformal lat
    xi = math.atan2(t, math.cos(dlon))
    eta = math.asinh(math.sin(dlon) / math.hypot(t, math.cos(dlon)))
    x, y = eta, xi
    for j, a in enumerate(al, start=1):
        x += a * math.cos(2 * j * xi) * math.sinh(2 * j * eta)
        y += a * math.sin(2 * j * xi) * math.cosh(2 * j * eta)
    k0 = 0.9996
    return 500000.0 + k0 * A * x, k0 * A * y


def _reference_offsets(reconstructions: List[Dict]) -> List[np.ndarray]:
    """(dx, dy, dalt) of every reconstruction's reference_lla relative to the first (:401-409, :432-444)."""
    lla0 = reconstructions[0].get("reference_lla")
    if lla0 is None:
        return [np.zeros(3) for _ in reconstructions]
    zone = utm_zone(lla0["longitude"])
    x0, y0 = utm_forward(lla0["longitude"], lla0["latitude"], zone)
    out = []
    for rec in reconstructions:
        lla = rec.get("reference_lla", lla0)
        x, y = utm_forward(lla["longitude"], lla["latitude"], zone)
        out.append(np.array([x - x0, y - y0, lla["altitude"] - lla0["altitude"]]))
    return out


# ----------------------------------------------------------------------------- readers
def read_opensfm(reconstructions: List[Dict]):
    """-> (cameras {id: Camera}, images {running index: Image})."""
    cameras: Dict[int, Camera] = {}
    ids_by_name: Dict[str, int] = {}
    images: Dict[int, Image] = {}
    next_cam = 1
    offsets = _reference_offsets(reconstructions)
    for rec, off in zip(reconstructions, offsets):
        for name, info in rec["cameras"].items():
            ptype = info["projection_type"]
            if ptype in ("spherical", "equirectangular"):
                cid = 0                                                      # every panorama camera shares id 0
                cameras[cid] = Camera(cid, "SPHERICAL", info["width"], info["height"], np.array([0]), True)
            elif ptype == "perspective":
                w, h = info["width"], info["height"]
                params = np.array([info["focal"] * w, w / 2, h / 2, info["k1"], info["k2"]])
                if name in ids_by_name:
                    cid = ids_by_name[name]
                else:
                    cid, next_cam = next_cam, next_cam + 1
                cameras[cid] = Camera(cid, "SIMPLE_PINHOLE", w, h, params, False)
            else:
                raise ValueError(f"camera {name!r}: projection_type {ptype!r} is not supported "
                                 "(perspective, spherical, equirectangular)")
            ids_by_name[name] = cid
        for shot_name, shot in rec["shots"].items():
            if shot["camera"] not in ids_by_name:
                raise KeyError(f"shot {shot_name!r} refers to unknown camera {shot['camera']!r}")
            idx = len(images)
            images[idx] = Image(id=idx, qvec=angle_axis_to_quaternion(shot["rotation"]),
                                tvec=np.asarray(shot["translation"], dtype=np.float64)[:3],
                                camera_id=ids_by_name[shot["camera"]], name=shot_name,
                                xys=np.array([0, 0]), point3D_ids=np.array([0, 0]), diff_ref=off)
    return cameras, images


def read_opensfm_points3D(reconstructions: List[Dict]):
    """-> (xyz [P,3] f64, rgb [P,3] f64 in 0..255, err [P,1] zeros).  Sign convention of :492:
    +dx, +dy, -dalt."""
    offsets = _reference_offsets(reconstructions)
    total = sum(len(rec["points"]) for rec in reconstructions)
    xyz = np.empty((total, 3))
    rgb = np.empty((total, 3))
    k = 0
    for rec, off in zip(reconstructions, offsets):
        shift = np.array([off[0], off[1], -off[2]])
        for pt in rec["points"].values():
            xyz[k] = np.asarray(pt["coordinates"], dtype=np.float64)[:3] + shift
            rgb[k] = pt["color"][:3]
            k += 1
    return xyz, rgb, np.zeros((total, 1))


# ----------------------------------------------------------------------------- undistortion
def _undistort_normalised(xd, yd, k1, k2, iters=5):
    """Inverse of the radial model by fixed-point iteration (OpenCV undistortPoints, 5 rounds)."""
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icd = 1.0 / (1.0 + (k2 * r2 + k1) * r2)
        icd = np.where(icd < 0, 1.0, icd)
        x, y = xd * icd, yd * icd
    return x, y


def undistort_maps(K, params, width, height):
    """(K_undist, roi [x,y,w,h], mapx, mapy) for dist = (k1,k2,p1,p2).
    getOptimalNewCameraMatrix(alpha=0): largest rectangle inscribed in the undistorted 9x9 border
    grid, in normalised coordinates -> new focal / principal point; then the sampling maps of
    initUndistortRectifyMap for that matrix.  With zero distortion this gives K back and the roi
    [0,0,width-1,height-1]: OpenCV's inscribed rectangle spans pixel centres 0..W-1, so the reference
    trains on images one pixel narrower and shorter than the files; kept, it decides H and W."""
    K = np.asarray(K, dtype=np.float64)
    k1, k2, p1, p2 = (float(v) for v in params)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    if p1 != 0.0 or p2 != 0.0:
        raise NotImplementedError("tangential distortion is never produced by the OpenSfM reader")

    def inner_rect(Knew):
        N = 9
        u = np.arange(N) * (width - 1) / (N - 1)
        v = np.arange(N) * (height - 1) / (N - 1)
        uu, vv = np.meshgrid(u, v, indexing="xy")
        x, y = _undistort_normalised((uu - cx) / fx, (vv - cy) / fy, k1, k2)
        if Knew is not None:
            x, y = x * Knew[0, 0] + Knew[0, 2], y * Knew[1, 1] + Knew[1, 2]
        x0, x1 = x[:, 0].max(), x[:, -1].min()
        y0, y1 = y[0, :].max(), y[-1, :].min()
        return x0, y0, x1 - x0, y1 - y0

    ix, iy, iw, ih = inner_rect(None)
    Kn = np.eye(3)
    Kn[0, 0], Kn[1, 1] = (width - 1) / iw, (height - 1) / ih
    Kn[0, 2], Kn[1, 2] = -Kn[0, 0] * ix, -Kn[1, 1] * iy
    rx, ry, rw, rh = (int(round(v)) for v in inner_rect(Kn))
    x_lo, y_lo = max(rx, 0), max(ry, 0)
    x_hi, y_hi = min(rx + rw, width), min(ry + rh, height)
    roi = [x_lo, y_lo, max(x_hi - x_lo, 0), max(y_hi - y_lo, 0)]

    gx, gy = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64), indexing="xy")
    x = (gx - Kn[0, 2]) / Kn[0, 0]
    y = (gy - Kn[1, 2]) / Kn[1, 1]
    r2 = x * x + y * y
    kr = 1.0 + (k2 * r2 + k1) * r2
    mapx = (fx * x * kr + cx).astype(np.float32)
    mapy = (fy * y * kr + cy).astype(np.float32)
    return Kn, roi, mapx, mapy


def remap_bilinear(image: np.ndarray, mapx: np.ndarray, mapy: np.ndarray) -> np.ndarray:
    """out[v,u] = bilinear sample of image at (mapx, mapy); zero outside (BORDER_CONSTANT)."""
    H, W = image.shape[:2]
    src = image.astype(np.float32)
    if src.ndim == 2:
        src = src[..., None]
    x0 = np.floor(mapx).astype(np.int64)
    y0 = np.floor(mapy).astype(np.int64)
    ax = (mapx - x0)[..., None]
    ay = (mapy - y0)[..., None]

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        return np.where(ok[..., None], src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)

    out = ((1 - ay) * ((1 - ax) * tap(y0, x0) + ax * tap(y0, x0 + 1))
           + ay * ((1 - ax) * tap(y0 + 1, x0) + ax * tap(y0 + 1, x0 + 1)))
    if image.ndim == 2:
        out = out[..., 0]
    if image.dtype == np.uint8:
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out.astype(image.dtype)


def _area_weights(n_in: int, n_out: int) -> np.ndarray:
    """[n_out, n_in] row-stochastic box-coverage weights of an area resample."""
    scale = n_in / n_out
    Wm = np.zeros((n_out, n_in))
    for o in range(n_out):
        lo, hi = o * scale, (o + 1) * scale
        for i in range(int(math.floor(lo)), min(int(math.ceil(hi)), n_in)):
            Wm[o, i] = max(0.0, min(hi, i + 1) - max(lo, i))
    return Wm / Wm.sum(axis=1, keepdims=True)


def resize_area(image: np.ndarray, new_size) -> np.ndarray:
    """Shrink to new_size=(width, height) by pixel-area averaging (cv2.INTER_AREA)."""
    nw, nh = new_size
    H, W = image.shape[:2]
    out = np.tensordot(_area_weights(H, nh), image.astype(np.float64), axes=(1, 0))
    out = np.moveaxis(np.tensordot(_area_weights(W, nw), out, axes=(1, 1)), 0, 1)
    if image.dtype == np.uint8:
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out.astype(image.dtype)


def read_image(path: str) -> np.ndarray:
    """[H,W,3] uint8.  `.npy` arrays are accepted next to the formats PIL decodes."""
    if path.endswith(".npy"):
        arr = np.load(path)
    else:
        from PIL import Image as PILImage
        with PILImage.open(path) as im:
            arr = np.asarray(im.convert("RGB"))
    if arr.ndim == 2:
        arr = np.repeat(arr[..., None], 3, axis=2)
    return np.ascontiguousarray(arr[..., :3])


# ----------------------------------------------------------------------------- Parser / Dataset
class Parser:
    """OpenSfM parser with the attribute surface of the reference's (opensfm.py:121-311)."""

    def __init__(self, data_dir: str, factor: int = 1, normalize: bool = False, test_every: int = 8):
        self.data_dir = data_dir
        self.factor = factor
        self.normalize = normalize
        self.test_every = test_every
        self._parse_reconstructions(self.load_reconstructions(data_dir))

    def load_reconstructions(self, data_dir):
        with open(os.path.join(data_dir, "reconstruction.json"), "r") as f:
            recs = json.load(f)
        if isinstance(recs, dict):
            recs = [recs]
        if not recs:
            raise ValueError("reconstruction.json holds no reconstruction")
        return recs

    def _parse_reconstructions(self, reconstructions: List[Dict]):
        self.cameras, self.images = read_opensfm(reconstructions)
        if not self.images:
            raise ValueError("reconstruction.json holds no shots")
        xyz, rgb, err = read_opensfm_points3D(reconstructions)
        points = xyz.astype(np.float32)
        self.points3D = xyz
        self.colors = rgb.astype(np.uint8)
        self.errors = err.astype(np.float32)

        f = self.factor
        w2c = np.tile(np.eye(4), (len(self.images), 1, 1))
        self.image_names, self.image_paths, self.camera_ids = [], [], []
        self.Ks_dict, self.params_dict, self.imsize_dict, self.mask_dict = {}, {}, {}, {}
        self.camtype_dict: Dict[int, str] = {}
        for i, img in self.images.items():
            w2c[i, :3, :3] = img.qvec2rotmat()
            w2c[i, :3, 3] = img.tvec
            self.image_names.append(img.name)
            self.image_paths.append(os.path.join(self.data_dir + "/images/", img.name))
            self.camera_ids.append(img.camera_id)
            cam = self.cameras[img.camera_id]
            if cam.model == "SIMPLE_PINHOLE":
                K = np.array([[cam.params[0], 0, cam.params[1]], [0, cam.params[0], cam.params[2]], [0, 0, 1.0]])
                K[:2, :] /= f
                self.Ks_dict[cam.id] = K
                self.params_dict[cam.id] = np.append(cam.params[3:5], [0.0, 0.0])   # (k1,k2,p1,p2)
                self.camtype_dict[cam.id] = "perspective"
            elif cam.model == "SPHERICAL":
                # the fork's own layout, not a pinhole K (:188); not divided by factor there either
                self.Ks_dict[cam.id] = np.array([[cam.width // 8, 0, 0], [0, cam.height // 4, 0],
                                                 [cam.width // 2, cam.height // 2, 1]])
                self.params_dict[cam.id] = np.empty(0, dtype=np.float32)
                self.camtype_dict[cam.id] = "spherical"
            self.imsize_dict[cam.id] = (cam.width // f, cam.height // f)
            self.mask_dict[cam.id] = None
        self.point_indices = {img.name: img.point3D_ids for img in self.images.values()}

        camtoworlds = np.linalg.inv(w2c)
        if self.normalize:
            T1 = similarity_from_cameras(camtoworlds)
            camtoworlds = transform_cameras(T1, camtoworlds)
            points = transform_points(T1, points)
            T2 = align_principle_axes(points)
            camtoworlds = transform_cameras(T2, camtoworlds)
            points = transform_points(T2, points)
            transform = T2 @ T1
        else:
            transform = np.eye(4)
        self.camtoworlds = camtoworlds
        self.points = points
        self.points_rgb = self.colors
        self.points_err = self.errors
        self.transform = transform

        self.mapx_dict, self.mapy_dict, self.roi_undist_dict = {}, {}, {}
        for cid, params in self.params_dict.items():
            if len(params) == 0:
                continue
            w, h = self.imsize_dict[cid]
            K_undist, roi, mapx, mapy = undistort_maps(self.Ks_dict[cid], params, w, h)
            self.mapx_dict[cid], self.mapy_dict[cid] = mapx, mapy
            self.Ks_dict[cid] = K_undist
            self.roi_undist_dict[cid] = roi
            self.imsize_dict[cid] = (roi[2], roi[3])
            self.mask_dict[cid] = None

        loc = camtoworlds[:, :3, 3]
        self.scene_scale = float(np.max(np.linalg.norm(loc - loc.mean(axis=0), axis=1)))

    def needs_undistort(self, camera_id: int) -> bool:
        p = self.params_dict[camera_id]
        return len(p) > 0 and bool(np.any(p != 0))


class Dataset:
    """Train/val view of a Parser; items are the dicts `Runner` consumes (opensfm.py:341-389)."""

    def __init__(self, parser: Parser, split: str = "train", patch_size: Optional[int] = None,
                 load_depths: bool = False):
        self.parser = parser
        self.split = split
        self.patch_size = patch_size
        self.load_depths = load_depths
        idx = np.arange(len(parser.images))
        if split == "train":
            idx = idx[idx % parser.test_every != 0]
        elif split == "val":
            idx = idx[idx % parser.test_every == 0]
        self.indices = idx
        self.image_name_to_local_idx = {parser.images[int(g)].name: l for l, g in enumerate(idx)}

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, item: int) -> Dict[str, Any]:
        p = self.parser
        g = int(self.indices[item])
        img = p.images[g]
        cid = img.camera_id
        K = np.array(p.Ks_dict[cid], dtype=np.float64)
        image = read_image(p.image_paths[g])
        if p.factor > 1:
            image = resize_area(image, (image.shape[1] // p.factor, image.shape[0] // p.factor))
        if len(p.params_dict[cid]) > 0:
            if p.needs_undistort(cid):
                image = remap_bilinear(image, p.mapx_dict[cid], p.mapy_dict[cid])
            x, y, w, h = p.roi_undist_dict[cid]
            image = image[y:y + h, x:x + w]
        if self.patch_size is not None:
            h, w = image.shape[:2]
            x = np.random.randint(0, max(w - self.patch_size, 1))
            y = np.random.randint(0, max(h - self.patch_size, 1))
            image = image[y:y + self.patch_size, x:x + self.patch_size]
            K[0, 2] -= x
            K[1, 2] -= y
        data = {
            "K": torch.from_numpy(K).float(),
            "camtoworld": torch.from_numpy(p.camtoworlds[g]).float(),
            "image": torch.from_numpy(np.require(image, requirements=["C", "W"])).float(),   # copies only a read-only decode
            "image_id": item,
            "image_name": img.name,
        }
        if self.load_depths:
            data["depths"] = torch.zeros(image.shape[:2], dtype=torch.float32)   # placeholder, as :384-386
        return data

    def get_data_by_image_name(self, image_name: str) -> Optional[Dict[str, Any]]:
        l = self.image_name_to_local_idx.get(image_name)
        return None if l is None else self[l]
