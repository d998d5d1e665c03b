"""Writes the scene-I/O fixtures under tests/golden/colmap_sparse/ byte by byte from COLMAP's documented binary model
format and the 3DGS .ply layout (real COLMAP / 3DGS tooling is not available offline; nothing here goes through
scene_io).  Values are chosen by hand so that tests/test_scene_io.py can state the expected numbers literally.

    python tests/golden/make_scene_fixtures.py
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
D = os.path.join(HERE, "colmap_sparse", "sparse", "0")
os.makedirs(D, exist_ok=True)

# cameras.bin: one SIMPLE_RADIAL camera (model id 2: f, cx, cy, k) -- what COLMAP's default feature extractor writes
with open(os.path.join(D, "cameras.bin"), "wb") as f:
    f.write(struct.pack("<Q", 1))
    f.write(struct.pack("<iiQQ", 7, 2, 1296, 840))
    f.write(struct.pack("<4d", 1040.5, 648.0, 420.0, 0.0125))

# images.bin: two images, written in the order (id 2, id 1); names sort the other way round
imgs = [
    (2, (0.5, 0.5, -0.5, 0.5), (0.25, -1.5, 3.0), 7, "frame_00002.JPG", [(10.5, 20.25, 11), (300.0, 400.0, -1)]),
    (1, (1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 4.0), 7, "frame_00001.JPG", [(648.0, 420.0, 11), (1.0, 2.0, 12), (3.0, 4.0, -1)]),
]
with open(os.path.join(D, "images.bin"), "wb") as f:
    f.write(struct.pack("<Q", len(imgs)))
    for iid, q, t, cid, name, pts in imgs:
        f.write(struct.pack("<i4d3di", iid, *q, *t, cid))
        f.write(name.encode() + b"\x00")
        f.write(struct.pack("<Q", len(pts)))
        for x, y, pid in pts:
            f.write(struct.pack("<ddq", x, y, pid))

# points3D.bin: two points with their tracks
pts3 = [(11, (0.0, 0.0, 0.0), (255, 128, 0), 0.75, [(1, 0), (2, 0)]), (12, (1.5, -2.0, 0.125), (1, 2, 3), 1.25, [(1, 1)])]
with open(os.path.join(D, "points3D.bin"), "wb") as f:
    f.write(struct.pack("<Q", len(pts3)))
    for pid, xyz, rgb, err, track in pts3:
        f.write(struct.pack("<Q3d3BdQ", pid, *xyz, *rgb, err, len(track)))
        for iid, idx in track:
            f.write(struct.pack("<ii", iid, idx))

# point_cloud.ply in the 3DGS layout (x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3), 3 vertices,
# value of property j of vertex i = i * 100 + j
names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
    ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
data = (np.arange(3)[:, None] * 100 + np.arange(len(names))[None, :]).astype("<f4")
with open(os.path.join(HERE, "colmap_sparse", "point_cloud.ply"), "wb") as f:
    f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 3\n")
    for nm in names:
        f.write(b"property float " + nm.encode() + b"\n")
    f.write(b"end_header\n")
    f.write(data.tobytes())
print("wrote", D)
