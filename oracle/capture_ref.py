"""capture_ref.py — CPU restatement of the capturer's voxelisation (TEST INFRASTRUCTURE).

Follows sender/capturer/capturer.py:88-126 step by step with numpy; Open3D's
`voxel_down_sample` (absent here) is restated from its published algorithm [RECALL]:
voxel index = floor((p - (min_bound - voxel_size/2)) / voxel_size) in double, position and colour
of a voxel = mean of its points in double, accumulated in input order.  PARITY UNPINNED against
Open3D itself (not installed).  The two places where the reference's result depends on an
unspecified order (hash-map iteration order of voxel_down_sample feeding np.unique's
first-occurrence rule; argpartition ties) are fixed the way the product documents:
duplicates keep the voxel with the smallest (ix,iy,iz); z ties go to the earlier (x,y,z) row.
"""
import numpy as np


def unpack_colors(xyzrgba):
    """capturer.py:91-94"""
    int_colors = np.ascontiguousarray(xyzrgba[:, 3]).view(np.uint32).reshape(-1, 1)
    return np.stack([((int_colors >> (8 * i)) & 0xFF) for i in range(3)], axis=-1).reshape(-1, 3)


def voxelize(xyzrgba, depth_clip=1.4, voxel_size=0.005, max_points=None):
    data = np.asarray(xyzrgba, dtype=np.float32)
    points = data[:, :3]
    colors = unpack_colors(data)
    # capturer.py:96-100
    with np.errstate(invalid="ignore", over="ignore"):
        distances = np.linalg.norm(points, axis=1)
        valid = np.isfinite(points).all(axis=1) & (distances <= np.float32(depth_clip))
    points, colors = points[valid], colors[valid]
    if points.shape[0] == 0:
        return {"points": np.zeros((0, 3), np.int16), "colors": np.zeros((0, 3), np.float64)}
    p64 = points.astype(np.float64)
    c64 = colors.astype(np.float64) / 255.0
    vs = np.float64(voxel_size)
    # Open3D voxel_down_sample [RECALL]
    vmb = p64.min(axis=0) - vs * 0.5
    idx = np.floor((p64 - vmb) / vs).astype(np.int64)
    key = (idx[:, 0] << 42) | (idx[:, 1] << 21) | idx[:, 2]
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    starts = np.flatnonzero(np.r_[True, key_s[1:] != key_s[:-1]])
    ends = np.r_[starts[1:], key_s.shape[0]]
    mean_p = np.empty((starts.shape[0], 3), np.float64)
    mean_c = np.empty((starts.shape[0], 3), np.float64)
    for v, (a, b) in enumerate(zip(starts, ends)):
        sp = np.zeros(3, np.float64)
        sc = np.zeros(3, np.float64)
        for j in order[a:b]:                     # input order inside the voxel
            sp = sp + p64[j]
            sc = sc + c64[j]
        mean_p[v] = sp / np.float64(b - a)
        mean_c[v] = sc / np.float64(b - a)
    # capturer.py:107: np.round(points / voxel_size).astype(np.int16)   (kept as int32 until the end)
    q = np.rint(mean_p / vs).astype(np.int64)
    # capturer.py:110-112: unique rows, first occurrence (here: in voxel-index order), sorted (x,y,z)
    lin = q[:, 0] * 10 ** 10 + q[:, 1] * 10 ** 5 + q[:, 2]
    o2 = np.argsort(lin, kind="stable")
    q, mean_c, lin = q[o2], mean_c[o2], lin[o2]
    first = np.r_[True, lin[1:] != lin[:-1]]
    q, mean_c = q[first], mean_c[first]
    # capturer.py:119-122: keep the max_points largest z
    if max_points is not None and q.shape[0] > max_points:
        rank = np.lexsort((np.arange(q.shape[0]), -q[:, 2]))      # z descending, then row ascending
        keep = np.sort(rank[:max_points])
        q, mean_c = q[keep], mean_c[keep]
    return {"points": q.astype(np.int16), "colors": mean_c}


def voxelize_open3d_semantics(xyzrgba, depth_clip=1.4, voxel_size=0.005):
    """dictionary form of the same steps (one accumulator per voxel index, like Open3D's
    unordered_map); returns {integer voxel: set of candidate colours} so that tests can check the
    restatement above on inputs where several Open3D voxels round to the same integer voxel."""
    data = np.asarray(xyzrgba, dtype=np.float32)
    points, colors = data[:, :3], unpack_colors(data)
    with np.errstate(invalid="ignore", over="ignore"):
        valid = np.isfinite(points).all(axis=1) & (np.linalg.norm(points, axis=1) <= np.float32(depth_clip))
    p64, c64 = points[valid].astype(np.float64), colors[valid].astype(np.float64) / 255.0
    vs = np.float64(voxel_size)
    vmb = p64.min(axis=0) - vs * 0.5
    acc = {}
    for p, c in zip(p64, c64):
        k = tuple(np.floor((p - vmb) / vs).astype(np.int64))
        if k not in acc:
            acc[k] = [np.zeros(3), np.zeros(3), 0]
        acc[k][0] = acc[k][0] + p
        acc[k][1] = acc[k][1] + c
        acc[k][2] += 1
    out = {}
    for sp, sc, n in acc.values():
        q = tuple(int(v) for v in np.rint((sp / n) / vs))
        out.setdefault(q, []).append(tuple(sc / n))
    return out
