"""Scene / camera I/O on either side of the hot path.

Counterparts of utils.load_checkpoint (utils.py:20-109), utils.get_viewmat_from_colmap_image (utils.py:215-219)
and the torch.save of the result (backproject.py:330,334).  `pycolmap_scene_manager` and `plyfile` are not
available offline, so the COLMAP binary model and the 3DGS .ply are parsed here with struct/numpy.
"""
from __future__ import annotations

import os
import struct
import warnings
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch

# COLMAP camera models: id -> (name, number of parameters)
_CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
                  4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
                  8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


@dataclass
class Camera:
    camera_id: int
    model: str
    width: int
    height: int
    params: np.ndarray

    @property
    def fx(self):
        return float(self.params[0])

    @property
    def fy(self):  # single-focal models share f
        return float(self.params[1] if self.model in ("PINHOLE", "OPENCV", "OPENCV_FISHEYE", "FULL_OPENCV",
                                                      "THIN_PRISM_FISHEYE") else self.params[0])

    @property
    def _two_focal(self):
        return self.model in ("PINHOLE", "OPENCV", "OPENCV_FISHEYE", "FULL_OPENCV", "THIN_PRISM_FISHEYE")

    @property
    def cx(self):
        return float(self.params[2] if self._two_focal else self.params[1])

    @property
    def cy(self):
        return float(self.params[3] if self._two_focal else self.params[2])


@dataclass
class Image:
    image_id: int
    qvec: np.ndarray  # (w, x, y, z), world -> camera
    tvec: np.ndarray
    camera_id: int
    name: str
    xys: Optional[np.ndarray] = None          # [P, 2] float64 2-D observations (images.bin)
    point3D_ids: Optional[np.ndarray] = None  # [P] int64, -1 = not triangulated

    @property
    def t(self):
        return self.tvec

    def R(self) -> np.ndarray:
        w, x, y, z = self.qvec / np.linalg.norm(self.qvec)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


@dataclass
class ColmapProject:
    """The subset of pycolmap_scene_manager.SceneManager the reference touches: .cameras, .images, and the sparse
    points its load_checkpoint also loads (utils.py:28-31: load_cameras, load_images, load_points3D)."""
    cameras: Dict[int, Camera] = field(default_factory=dict)
    images: Dict[int, Image] = field(default_factory=dict)
    points3D: Optional[np.ndarray] = None        # [M, 3] float64
    point3D_ids: Optional[np.ndarray] = None     # [M] uint64
    point3D_colors: Optional[np.ndarray] = None  # [M, 3] uint8
    point3D_errors: Optional[np.ndarray] = None  # [M] float64
    point3D_id_to_images: Dict[int, np.ndarray] = field(default_factory=dict)  # id -> [T, 2] (image_id, point2D_idx)


def read_colmap_model(sparse_dir: str) -> ColmapProject:
    """cameras.bin + images.bin of a COLMAP sparse model (utils.py:28-31 loads the same through pycolmap)."""
    proj = ColmapProject()
    with open(os.path.join(sparse_dir, "cameras.bin"), "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            cid, mid, w, h = struct.unpack("<iiQQ", f.read(24))
            if mid not in _CAMERA_MODELS:
                raise ValueError(f"unknown COLMAP camera model id {mid}")
            name, npar = _CAMERA_MODELS[mid]
            params = np.array(struct.unpack("<" + "d" * npar, f.read(8 * npar)))
            proj.cameras[cid] = Camera(cid, name, int(w), int(h), params)
    with open(os.path.join(sparse_dir, "images.bin"), "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            iid = struct.unpack("<i", f.read(4))[0]
            q = np.array(struct.unpack("<4d", f.read(32)))
            t = np.array(struct.unpack("<3d", f.read(24)))
            cid = struct.unpack("<i", f.read(4))[0]
            name = bytearray()
            while True:
                c = f.read(1)
                if c in (b"\x00", b""):
                    break
                name += c
            (npts,) = struct.unpack("<Q", f.read(8))
            obs = np.frombuffer(f.read(24 * npts), dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")]))
            proj.images[iid] = Image(iid, q, t, cid, name.decode("utf-8"),
                                     np.stack([obs["x"], obs["y"]], axis=1) if npts else np.zeros((0, 2)),
                                     obs["id"].copy())
    p3 = os.path.join(sparse_dir, "points3D.bin")
    if os.path.exists(p3):  # utils.py:31 load_points3D (the back-projection itself never reads them)
        read_points3D(p3, proj)
    return proj


def read_points3D(path: str, proj: Optional[ColmapProject] = None) -> ColmapProject:
    """points3D.bin of a COLMAP sparse model: per point id (u64), xyz (3 x f64), rgb (3 x u8), reprojection error
    (f64) and its track, (image_id i32, point2D_idx i32) per observation."""
    proj = proj if proj is not None else ColmapProject()
    ids, xyz, rgb, err = [], [], [], []
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            pid, x, y, z, r, g, b, e, tl = struct.unpack("<Q3d3BdQ", f.read(8 + 24 + 3 + 8 + 8))
            track = np.frombuffer(f.read(8 * tl), dtype="<i4").reshape(tl, 2)
            ids.append(pid), xyz.append((x, y, z)), rgb.append((r, g, b)), err.append(e)
            proj.point3D_id_to_images[int(pid)] = track.copy()
    proj.point3D_ids = np.asarray(ids, dtype=np.uint64)
    proj.points3D = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    proj.point3D_colors = np.asarray(rgb, dtype=np.uint8).reshape(-1, 3)
    proj.point3D_errors = np.asarray(err, dtype=np.float64)
    return proj


def camera_matrix(cam: Camera, data_factor: int = 1) -> torch.Tensor:
    """utils.py:93-105: K from the first camera, K[:2,:3] /= data_factor."""
    K = torch.tensor([[cam.fx, 0.0, cam.cx], [0.0, cam.fy, cam.cy], [0.0, 0.0, 1.0]])
    K[:2, :3] /= data_factor
    return K


def get_viewmat_from_colmap_image(image: Image) -> torch.Tensor:
    """utils.py:215-219: [R t; 0 0 0 1] world -> camera."""
    vm = torch.eye(4).float()
    vm[:3, :3] = torch.tensor(image.R()).float()
    vm[:3, 3] = torch.tensor(image.t).float()
    return vm


def read_gaussian_ply(path: str) -> Dict[str, torch.Tensor]:
    """3DGS (inria) .ply -> the reference's splat keys (utils.py:68-85), including its reshape of f_rest_*
    to (-1, 15, 3) exactly as written there."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, props, count = None, [], 0
        in_vertex = False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii").split()
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties are not supported in the vertex element")
                props.append((tok[2], {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8",
                                       "uchar": "u1", "uint8": "u1", "int": "<i4", "int32": "<i4",
                                       "uint": "<u4", "short": "<i2", "ushort": "<u2", "char": "i1"}[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError(f"unsupported PLY format {fmt}")
        v = np.frombuffer(f.read(count * np.dtype(props).itemsize), dtype=np.dtype(props), count=count)

    def col(names):
        return torch.from_numpy(np.stack([v[n].astype(np.float32) for n in names], axis=1))

    return {
        "active_sh_degree": 3,
        "means": col(["x", "y", "z"]),
        "features_dc": col(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(-1, 1, 3),
        "features_rest": col([f"f_rest_{i}" for i in range(45)]).reshape(-1, 15, 3),
        "scaling": col([f"scale_{i}" for i in range(3)]),
        "rotation": col([f"rot_{i}" for i in range(4)]),
        "opacity": col(["opacity"])[:, 0],
    }


def load_checkpoint(checkpoint: str, data_dir: str, format: Optional[str] = "gsplat", data_factor: int = 1,
                    rasterizer: Optional[str] = None) -> Dict:
    """Counterpart of utils.load_checkpoint (utils.py:20-109): same formats, same keys, same K handling."""
    colmap_project = read_colmap_model(os.path.join(data_dir, "sparse", "0"))
    if format is None and rasterizer is None:
        raise ValueError("Must specify format or rasterizer")
    if rasterizer is not None:
        format = rasterizer
        warnings.warn("`rasterizer` is deprecated. Use `format` instead.", DeprecationWarning)
    if format in ("inria", "gsplat"):
        model = torch.load(checkpoint, weights_only=False, map_location="cpu")
    if format == "inria":
        p, _ = model
        splats = {"active_sh_degree": p[0], "means": p[1], "features_dc": p[2], "features_rest": p[3],
                  "scaling": p[4], "rotation": p[5], "opacity": p[6].squeeze(1)}
    elif format == "gsplat":
        p = model["splats"]
        splats = {"active_sh_degree": 3, "means": p["means"], "features_dc": p["sh0"], "features_rest": p["shN"],
                  "scaling": p["scales"], "rotation": p["quats"], "opacity": p["opacities"]}
    elif format == "ply":
        splats = read_gaussian_ply(checkpoint)
    else:
        raise ValueError("Invalid Gaussian splatting format")
    for k, val in splats.items():
        if isinstance(val, torch.Tensor):
            splats[k] = val.detach()
    cam = next(iter(colmap_project.cameras.values()))  # "Assuming only one camera" (utils.py:92)
    splats["camera_matrix"] = camera_matrix(cam, data_factor)
    splats["colmap_project"] = colmap_project
    splats["colmap_dir"] = data_dir
    return splats


def sorted_viewmats(colmap_project: ColmapProject) -> torch.Tensor:
    """[V,4,4] in the reference's iteration order: images sorted by name (backproject.py:74)."""
    imgs = sorted(colmap_project.images.values(), key=lambda im: im.name)
    return torch.stack([get_viewmat_from_colmap_image(im) for im in imgs])


def save_features(features: torch.Tensor, results_dir: str, name: str = "features_lseg.pt") -> str:
    """backproject.py:330 / :334 / backproject_compressed.py:217."""
    os.makedirs(results_dir, exist_ok=True)
    path = os.path.join(results_dir, name)
    torch.save(features, path)
    return path
