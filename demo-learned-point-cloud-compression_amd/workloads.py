"""Seeded synthetic inputs for the configurations of BASELINE.json.

Frames follow the reference's input schema (sender/capturer/capturer.py:111-126):
{"points": int16 [N,3] unique voxel coordinates (may be negative),
 "colors": float64 [N,3] in [0,1]}.  numpy only.
"""
import numpy as np


def _dedup(pts):
    """unique rows in lexicographic (x, y, z) order — np.unique(axis=0)'s result, through packed 63-bit keys"""
    p = pts.astype(np.int64)
    bias, mask = 1 << 20, (1 << 21) - 1
    k = np.unique(((p[:, 0] + bias) << 42) | ((p[:, 1] + bias) << 21) | (p[:, 2] + bias))
    return np.stack([(k >> 42) - bias, ((k >> 21) & mask) - bias, (k & mask) - bias], 1)


def _colors(pts, rng, noise=0.02):
    p = pts.astype(np.float64)
    c = np.stack([0.5 + 0.5 * np.sin(p[:, 0] * 0.05 + 0.3 * np.cos(p[:, 1] * 0.03)),
                  0.5 + 0.5 * np.sin(p[:, 1] * 0.04 + 1.0),
                  0.5 + 0.5 * np.sin(p[:, 2] * 0.06 + p[:, 0] * 0.01 + 2.0)], axis=1)
    c += rng.normal(0, noise, c.shape)
    return np.clip(c, 0.0, 1.0)


def frame(pts, rng, offset=(0, 0, 0)):
    pts = _dedup(pts) + np.asarray(offset, dtype=np.int64)
    return {"points": pts.astype(np.int16), "colors": _colors(pts, rng)}


def sphere_shell(grid=64, radius=25.2, seed=0, offset=(0, 0, 0)):
    """C1: voxels of a grid^3 cube with |norm(v - c) - R| < 0.5 (~8k voxels at 64^3)"""
    rng = np.random.default_rng(seed)
    g = np.arange(grid)
    v = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    d = np.linalg.norm(v - (grid - 1) / 2.0, axis=1)
    return frame(v[np.abs(d - radius) < 0.5], rng, offset)


def _plane(rng, extent, axis, level, n):
    p = np.empty((n, 3), dtype=np.int64)
    others = [a for a in range(3) if a != axis]
    for a in others:
        p[:, a] = rng.integers(0, extent[a], n)
    p[:, axis] = level
    return p


def _box_surface(rng, lo, size, n):
    p = rng.integers(0, 1 << 30, (n, 3)) % np.asarray(size) + np.asarray(lo)
    face = rng.integers(0, 6, n)
    for a in range(3):
        p[face == 2 * a, a] = lo[a]
        p[face == 2 * a + 1, a] = lo[a] + size[a] - 1
    return p


def _sphere_surface(rng, centre, r, n):
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return np.rint(v * r + np.asarray(centre)).astype(np.int64)


def room(n_target=1_000_000, extent=(512, 512, 256), seed=0, offset=(-200, -150, -100)):
    """C2: ScanNet-scale indoor-like frame: 6 walls/floor/ceiling planes, 20
    boxes, 10 spheres inside a 512x512x256 grid; sub-sampled to exactly
    n_target unique voxels (fewer if the scene has fewer)."""
    rng = np.random.default_rng(seed)
    ex = np.asarray(extent)
    parts = []
    # dense planes (all voxels of the six faces)
    for axis in range(3):
        o = [a for a in range(3) if a != axis]
        g = np.stack(np.meshgrid(np.arange(ex[o[0]]), np.arange(ex[o[1]]), indexing="ij"), -1).reshape(-1, 2)
        for level in (0, ex[axis] - 1):
            p = np.empty((g.shape[0], 3), dtype=np.int64)
            p[:, o[0]], p[:, o[1]], p[:, axis] = g[:, 0], g[:, 1], level
            parts.append(p)
    sc = min(1.0, float(ex.min()) / 256.0)     # furniture shrinks with rooms smaller than the default (tests)
    for _ in range(20):
        size = rng.integers(max(2, int(20 * sc)), max(4, int(120 * sc)), 3)
        lo = rng.integers(1, ex - size - 1)
        area = 2 * (size[0] * size[1] + size[1] * size[2] + size[0] * size[2])
        parts.append(_box_surface(rng, lo, size, int(area * 3)))
    for _ in range(10):
        r = int(rng.integers(max(2, int(15 * sc)), max(4, int(50 * sc))))
        c = rng.integers(r + 1, ex - r - 1)
        parts.append(_sphere_surface(rng, c, r, int(4 * np.pi * r * r * 4)))
    pts = _dedup(np.concatenate(parts, 0))
    pts = pts[(pts >= 0).all(1) & (pts < ex).all(1)]
    if pts.shape[0] > n_target:
        # keep a spatially coherent subset: drop whole z-slabs from the top until close, then trim randomly
        keep = rng.permutation(pts.shape[0])[:n_target]
        pts = pts[np.sort(keep)]
    return frame(pts, rng, offset)


def lidar_sweep(n_beams=64, n_az=1800, seed=0, voxel=0.02):
    """C3: KITTI-like spinning LiDAR (geometry-only use): ground plane + boxes,
    quantised to `voxel` metres, de-duplicated (~120k voxels)."""
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))
    az = np.linspace(0, 2 * np.pi, n_az, endpoint=False)
    el, az = np.meshgrid(el, az, indexing="ij")
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1).reshape(-1, 3)
    h = 1.73
    t = np.full(d.shape[0], np.inf)
    down = d[:, 2] < -1e-3
    t[down] = -h / d[down, 2]
    for _ in range(30):
        c = np.array([rng.uniform(-40, 40), rng.uniform(-40, 40), rng.uniform(-h, 0.5)])
        half = np.array([rng.uniform(0.5, 4), rng.uniform(0.5, 4), rng.uniform(0.5, 2)])
        lo, hi = c - half, c + half
        with np.errstate(divide="ignore", invalid="ignore"):
            t1, t2 = lo / d, hi / d
        tn, tf = np.minimum(t1, t2).max(1), np.maximum(t1, t2).min(1)
        hit = (tn < tf) & (tn > 0.5)
        t = np.where(hit & (tn < t), tn, t)
    ok = np.isfinite(t) & (t < 80)
    p = d[ok] * t[ok, None] + rng.normal(0, 0.005, (ok.sum(), 3))
    pts = np.rint(p / voxel).astype(np.int64)
    pts = pts[(np.abs(pts) < 32000).all(1)]
    return frame(pts, rng)


def body(n_target=800_000, bits=10, seed=0):
    """C4: 8iVFB-like dense closed surface on a 2^bits grid (union of ellipsoids), RGB texture"""
    rng = np.random.default_rng(seed)
    size = 1 << bits
    parts = []
    ell = [((0.5, 0.5, 0.45), (0.17, 0.11, 0.30)), ((0.5, 0.5, 0.83), (0.09, 0.09, 0.10)),
           ((0.36, 0.5, 0.5), (0.06, 0.06, 0.26)), ((0.64, 0.5, 0.5), (0.06, 0.06, 0.26)),
           ((0.44, 0.5, 0.16), (0.07, 0.07, 0.16)), ((0.56, 0.5, 0.16), (0.07, 0.07, 0.16))]
    for c, r in ell:
        c, r = np.asarray(c) * size, np.asarray(r) * size
        n = int(4 * np.pi * (r[0] * r[1] + r[1] * r[2] + r[0] * r[2]) / 3 * 5)
        v = rng.normal(size=(n, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        parts.append(np.rint(v * r + c).astype(np.int64))
    pts = _dedup(np.concatenate(parts, 0))
    # remove interior points (inside another ellipsoid)
    inside = np.zeros(pts.shape[0], dtype=bool)
    for c, r in ell:
        c, r = np.asarray(c) * size, np.asarray(r) * size
        inside |= (((pts - c) / (r - 1.5)) ** 2).sum(1) < 1.0
    pts = pts[~inside]
    if pts.shape[0] > n_target:
        pts = pts[np.sort(rng.permutation(pts.shape[0])[:n_target])]
    return frame(pts, rng, offset=(-size // 2, -size // 2, -size // 2))


def tiled_block(t, n_per_tile=500_000, seed=0):
    """block t of the C5 scan: an independent C2-style room (its own seed, a slightly shifted origin)"""
    off = (-200 + 40 * (t & 1), -150 + 30 * ((t >> 1) & 1), -100 + 20 * ((t >> 2) & 1))
    return room(n_per_tile, seed=seed + 1 + t, offset=off)


def tiled_scan(tiles=8, n_per_tile=500_000, seed=0):
    """C5: 2x2x2 spatial tiles, each an independent C2-style block = one batch item"""
    return [tiled_block(t, n_per_tile, seed) for t in range(tiles)]


def fused_scan(n_total=4_000_000, seed=0, extent=(512, 512, 256)):
    """C5 as ONE frame: a fused indoor scan of 2x2x2 C2-style rooms side by side (1024 x 1024 x 512 voxels at the
    default extent, origin at the centre), n_total / 8 voxels per room; `tiled.cut_tiles` with blocks of one room
    cuts it back into the 8 octree blocks"""
    ex = np.asarray(extent)
    pts, cols = [], []
    for t in range(8):
        o = np.array([(t >> 2) & 1, (t >> 1) & 1, t & 1]) * ex - ex
        f = room(n_total // 8, extent=extent, seed=seed + 101 + t, offset=tuple(int(v) for v in o))
        pts.append(f["points"])
        cols.append(f["colors"])
    return {"points": np.concatenate(pts, 0), "colors": np.concatenate(cols, 0)}


def gop(frames):
    """wrap frames the way sender/encoder/encoder.py:123-137 hands them to compress()"""
    return {"frames": frames, "timestamps": {"capturing": [0.0] * len(frames), "sampling": 0.0},
            "segment_duration": 1.0, "frame_rate": len(frames)}
