"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's post-path geometry helper.

depth_to_world_coords_points follows src/models/utils/geometry.py:5-55 (depth_to_camera_coords) and :57-89
(camera-to-world transform), in numpy fp32 with the reference's operation order.  Pinned against outputs of the
reference function itself (tests/golden/geometry_depth_to_world.npz, written by oracle/gen_golden_geometry.py).
Only tests/ may import this module."""
import numpy as np


def depth_to_world_coords_points(depth_map, extrinsic, intrinsic, eps=1e-8):
    depth_map = np.asarray(depth_map, np.float32)
    extrinsic = np.asarray(extrinsic, np.float32)
    intrinsic = np.asarray(intrinsic, np.float32)
    B, H, W = depth_map.shape
    point_mask = depth_map > eps                                            # geometry.py:76
    fx, fy = intrinsic[:, 0, 0], intrinsic[:, 1, 1]                         # :24-27
    cx, cy = intrinsic[:, 0, 2], intrinsic[:, 1, 2]
    v, u = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")  # :30-34
    z = depth_map
    x = (u[None] - cx[:, None, None]) * z / fx[:, None, None]               # :45
    y = (v[None] - cy[:, None, None]) * z / fy[:, None, None]               # :46
    cam = np.stack([x, y, z], -1).astype(np.float32)                        # :49
    R, t = extrinsic[:, :3, :3], extrinsic[:, :3, 3]                        # :82-83
    world = np.einsum("bhwi,bji->bhwj", cam, R).astype(np.float32) + t[:, None, None, :]  # :86
    return world.astype(np.float32), cam, point_mask


def create_confidence_mask(confidence, conf_threshold_percent=30.0):
    """infer.py:25-59 restated in numpy.  Ties at the K-th value: lowest flat index first (stable sort) — the reference's
    torch.topk leaves that choice unspecified, so fixtures are tie-free at the threshold."""
    c = np.asarray(confidence, np.float32).reshape(-1).copy()
    c[c <= 1e-5] = -np.inf                                   # :40
    n = c.size
    k = int(np.ceil(n * (100.0 - conf_threshold_percent) / 100.0)) if conf_threshold_percent > 0 else n   # :44-48
    k = max(1, k)                                            # :49
    order = np.argsort(-c, kind="stable")                    # top-k, lowest index first among equals (:52)
    mask = np.zeros(n, bool)
    mask[order[:k]] = True                                   # :55-56
    return mask

