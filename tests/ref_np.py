"""Independent float64 numpy formulation of SURVEY.md section 3.3 (test helper, small scenes only).

Written from the maths (matrix form), NOT from the fp32 operation order of the oracle / HIP kernels:
    Sigma = R_q diag(s^2) R_q^T,  Sigma_c = R Sigma R^T,  Sigma_2 = J Sigma_c J^T + eps2d I,  conic = Sigma_2^-1
    per pixel: Gaussians whose tile rectangle contains the pixel's tile, ascending (depth, index),
    alpha = min(0.999, o exp(-sigma)); skip alpha < 1/255; stop (uncounted) when T (1 - alpha) <= 1e-4.
Returns the dense weight matrix W[P, N] (row-major pixels) so that F = W^T feats, d = W^T 1.
"""
import numpy as np


def quat_to_rot(q):
    q = q / np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def project(means, quats, scales, viewmat, K, W, H, near=0.01, far=1e10, eps2d=0.3, tile=16):
    means, quats, scales = (np.asarray(a, np.float64) for a in (means, quats, scales))
    vm, K = np.asarray(viewmat, np.float64), np.asarray(K, np.float64)
    R, t = vm[:3, :3], vm[:3, 3]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    n = means.shape[0]
    tw, th = -(-W // tile), -(-H // tile)
    out = dict(ok=np.zeros(n, bool), mu=np.zeros((n, 2)), conic=np.zeros((n, 3)), z=np.zeros(n),
               radius=np.zeros(n, int), rect=np.zeros((n, 4), int))
    limxp, limxn = (W - cx) / fx + 0.3 * 0.5 * W / fx, cx / fx + 0.3 * 0.5 * W / fx
    limyp, limyn = (H - cy) / fy + 0.3 * 0.5 * H / fy, cy / fy + 0.3 * 0.5 * H / fy
    for i in range(n):
        mc = R @ means[i] + t
        if mc[2] < near or mc[2] > far:
            continue
        Rq = quat_to_rot(quats[i])
        S = Rq @ np.diag(scales[i] ** 2) @ Rq.T
        Sc = R @ S @ R.T
        x, y, z = mc
        tx = z * min(limxp, max(-limxn, x / z))
        ty = z * min(limyp, max(-limyn, y / z))
        J = np.array([[fx / z, 0, -fx * tx / z ** 2], [0, fy / z, -fy * ty / z ** 2]])
        S2 = J @ Sc @ J.T + eps2d * np.eye(2)
        det = np.linalg.det(S2)
        if det <= 0:
            continue
        b = 0.5 * (S2[0, 0] + S2[1, 1])
        radius = int(np.ceil(3 * np.sqrt(b + np.sqrt(max(0.01, b * b - det)))))
        u, v = fx * x / z + cx, fy * y / z + cy
        if radius <= 0 or u + radius <= 0 or u - radius >= W or v + radius <= 0 or v - radius >= H:
            continue
        Ci = np.linalg.inv(S2)
        out["ok"][i], out["mu"][i], out["z"][i], out["radius"][i] = True, (u, v), z, radius
        out["conic"][i] = (Ci[0, 0], Ci[0, 1], Ci[1, 1])
        out["rect"][i] = (min(max(int(np.floor((u - radius) / tile)), 0), tw),
                          min(max(int(np.floor((v - radius) / tile)), 0), th),
                          min(max(int(np.ceil((u + radius) / tile)), 0), tw),
                          min(max(int(np.ceil((v + radius) / tile)), 0), th))
    return out


def weights(proj, opac, W, H, tile=16):
    n = proj["ok"].shape[0]
    Wm = np.zeros((H * W, n))
    order = sorted(np.nonzero(proj["ok"])[0], key=lambda i: (np.float32(proj["z"][i]), i))
    alpha_map = np.zeros((H, W))
    for iy in range(H):
        for ix in range(W):
            tx, ty = ix // tile, iy // tile
            T = 1.0
            for i in order:
                x0, y0, x1, y1 = proj["rect"][i]
                if not (x0 <= tx < x1 and y0 <= ty < y1):
                    continue
                dx, dy = proj["mu"][i][0] - (ix + 0.5), proj["mu"][i][1] - (iy + 0.5)
                a, b, c = proj["conic"][i]
                sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
                alpha = min(0.999, opac[i] * np.exp(-sigma))
                if sigma < 0 or alpha < 1 / 255:
                    continue
                nT = T * (1 - alpha)
                if nT <= 1e-4:
                    break
                Wm[iy * W + ix, i] = alpha * T
                T = nT
            alpha_map[iy, ix] = 1 - T
    return Wm, alpha_map
