"""N2: COLMAP binary model / checkpoint / .ply readers round-trip (no GPU)."""
import os
import struct

import numpy as np
import torch

import gsbp_amd
from gsbp_amd import scene_io as sio
from gsbp_amd import synthetic as syn


def _write_colmap(d, K, vms, names, w, h):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", 1))
        f.write(struct.pack("<iiQQ", 1, 1, w, h))  # PINHOLE
        f.write(struct.pack("<4d", K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
    with open(os.path.join(d, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(names)))
        for i, (vm, name) in enumerate(zip(vms, names)):
            R = vm[:3, :3].astype(np.float64)
            qw = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
            q = np.array([qw, (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)])
            f.write(struct.pack("<i", i + 1))
            f.write(struct.pack("<4d", *q))
            f.write(struct.pack("<3d", *vm[:3, 3].astype(np.float64)))
            f.write(struct.pack("<i", 1))
            f.write(name.encode() + b"\x00")
            f.write(struct.pack("<Q", 2))
            f.write(struct.pack("<ddq", 1.0, 2.0, -1) * 2)


def test_colmap_and_checkpoint_roundtrip(tmp_path):
    cfg = syn.CONFIGS["T0"]
    K = syn.intrinsics(cfg).numpy().astype(np.float64)
    vms = syn.make_cameras(cfg, n_views=3).numpy()
    names = ["b.png", "a.png", "c.png"]
    data_dir = str(tmp_path / "scene")
    _write_colmap(os.path.join(data_dir, "sparse", "0"), K * np.array([[2, 1, 2], [1, 2, 2], [1, 1, 1.0]]), vms, names,
                  2 * cfg.width, 2 * cfg.height)
    sc = syn.make_scene(cfg)
    n = cfg.n_gaussians
    ck = {"splats": {"means": sc["means"], "sh0": torch.zeros(n, 1, 3), "shN": torch.zeros(n, 15, 3),
                     "scales": sc["scaling"], "quats": sc["rotation"], "opacities": sc["opacity"]}}
    path = str(tmp_path / "ckpt.pt")
    torch.save(ck, path)
    splats = sio.load_checkpoint(path, data_dir, format="gsplat", data_factor=2)
    assert set(splats) >= {"means", "features_dc", "features_rest", "scaling", "rotation", "opacity", "camera_matrix",
                           "colmap_project", "colmap_dir", "active_sh_degree"}
    assert torch.allclose(splats["camera_matrix"], syn.intrinsics(cfg), atol=1e-6)  # K[:2,:3] /= data_factor
    assert int(splats["camera_matrix"][0, 2] * 2) == cfg.width  # backproject.py:85
    vm_sorted = sio.sorted_viewmats(splats["colmap_project"])  # sorted by name: a, b, c
    assert torch.allclose(vm_sorted[0], torch.from_numpy(vms[1]), atol=1e-5)
    assert torch.allclose(vm_sorted[1], torch.from_numpy(vms[0]), atol=1e-5)
    # inria tuple format
    inria = ((3, sc["means"], torch.zeros(n, 1, 3), torch.zeros(n, 15, 3), sc["scaling"], sc["rotation"],
              sc["opacity"][:, None]), 0)
    torch.save(inria, path)
    s2 = sio.load_checkpoint(path, data_dir, format=None, rasterizer="inria")
    assert torch.equal(s2["opacity"], sc["opacity"]) and s2["active_sh_degree"] == 3
    assert sio.save_features(torch.zeros(3, 4), str(tmp_path / "res")).endswith("features_lseg.pt")


def test_ply_reader(tmp_path):
    n = 7
    rng = np.random.default_rng(0)
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    data = rng.standard_normal((n, len(names))).astype("<f4")
    path = str(tmp_path / "pc.ply")
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n)
        for nm in names:
            f.write(b"property float " + nm.encode() + b"\n")
        f.write(b"end_header\n")
        f.write(data.tobytes())
    s = sio.read_gaussian_ply(path)
    assert s["means"].shape == (n, 3) and s["features_rest"].shape == (n, 15, 3) and s["rotation"].shape == (n, 4)
    np.testing.assert_array_equal(s["means"].numpy(), data[:, :3])
    np.testing.assert_array_equal(s["features_rest"].numpy().reshape(n, 45), data[:, 9:54])  # utils.py:79-81 reshape
    np.testing.assert_array_equal(s["opacity"].numpy(), data[:, 54])


def test_committed_colmap_model_and_ply_fixture():
    """Known-answer test on the committed bytes of tests/golden/colmap_sparse (written by make_scene_fixtures.py straight
    from COLMAP's binary-model format: SIMPLE_RADIAL camera, images out of name order with 2-D observations,
    points3D.bin with tracks; a .ply in the 3DGS property order incl. normals): utils.load_checkpoint's behaviour
    (utils.py:20-109) and get_viewmat_from_colmap_image (utils.py:215-219)."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colmap_sparse")
    s = sio.load_checkpoint(os.path.join(root, "point_cloud.ply"), root, format="ply", data_factor=2)
    proj = s["colmap_project"]
    cam = proj.cameras[7]
    assert (cam.model, cam.width, cam.height) == ("SIMPLE_RADIAL", 1296, 840)
    assert (cam.fx, cam.fy, cam.cx, cam.cy) == (1040.5, 1040.5, 648.0, 420.0)  # single focal length shared by x and y
    assert torch.equal(s["camera_matrix"], torch.tensor([[520.25, 0.0, 324.0], [0.0, 520.25, 210.0], [0.0, 0.0, 1.0]]))
    assert int(s["camera_matrix"][0, 2] * 2) == 648 and int(s["camera_matrix"][1, 2] * 2) == 420  # backproject.py:85-86
    assert [im.name for im in sorted(proj.images.values(), key=lambda x: x.name)] == ["frame_00001.JPG", "frame_00002.JPG"]
    vms = sio.sorted_viewmats(proj)
    assert torch.equal(vms[0], torch.tensor([[1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, 1.0, 4.0], [0, 0, 0, 1.0]]))
    # q = (w, x, y, z) = (.5, .5, -.5, .5): R = [[0, -1, 0], [0, 0, -1], [1, 0, 0]]
    assert torch.allclose(vms[1], torch.tensor([[0.0, -1, 0, 0.25], [0, 0, -1.0, -1.5], [1.0, 0, 0, 3.0], [0, 0, 0, 1.0]]))
    im1 = proj.images[1]
    assert im1.xys.shape == (3, 2) and im1.point3D_ids.tolist() == [11, 12, -1] and im1.xys[0].tolist() == [648.0, 420.0]
    assert proj.point3D_ids.tolist() == [11, 12] and proj.points3D[1].tolist() == [1.5, -2.0, 0.125]
    assert proj.point3D_colors[0].tolist() == [255, 128, 0] and proj.point3D_errors.tolist() == [0.75, 1.25]
    assert proj.point3D_id_to_images[11].tolist() == [[1, 0], [2, 0]]
    # .ply: property j of vertex i holds i * 100 + j; f_rest_* reshaped to (-1, 15, 3) as utils.py:79-81 writes it
    assert s["means"].tolist() == [[0.0, 1.0, 2.0], [100.0, 101.0, 102.0], [200.0, 201.0, 202.0]]
    assert s["features_dc"][1, 0].tolist() == [106.0, 107.0, 108.0]
    assert s["features_rest"][2, 0].tolist() == [209.0, 210.0, 211.0] and s["features_rest"][2, 14, 2].item() == 253.0
    assert s["opacity"].tolist() == [54.0, 154.0, 254.0]
    assert s["scaling"][0].tolist() == [55.0, 56.0, 57.0] and s["rotation"][0].tolist() == [58.0, 59.0, 60.0, 61.0]
