"""Rate / distortion figures of BASELINE.json's metric ("bpp & D1-PSNR").

The reference logs only rates — bpp = 8 * len(container) / N, raw = 48 bpp
(sender/encoder/codec_pipeline.py:226-230) — and has no distortion measure; SURVEY.md §8(d)
fixes the definition used here, MPEG's `pc_error` point-to-point (D1) metric:

    e(A -> B)   = mean over a in A of |a - nn_B(a)|^2            (nearest neighbour in B)
    D1-PSNR     = 10 log10( 3 p^2 / max(e(A -> B), e(B -> A)) )   p = peak = grid extent - 1
    Y-PSNR      = 10 log10( 1 / max(c(A -> B), c(B -> A)) )       c = mean squared difference of the BT.709
                  luma (colours in [0, 1]) between a point and its nearest neighbour in the other cloud

Measurement code (host, float64, scipy's k-d tree); not on the codec's path.
"""
import numpy as np

BT709 = np.array([0.2126, 0.7152, 0.0722], dtype=np.float64)


def _nn(a, b):
    """for every row of a: squared distance to and index of its nearest neighbour among the rows of b"""
    from scipy.spatial import cKDTree
    from ._abi import host_cpu_budget          # the cgroup's CPU share, not the visible core count
    d, i = cKDTree(b).query(a, k=1, workers=max(1, host_cpu_budget()))
    return d * d, i


def peak_of(points):
    """p = grid extent - 1 of a synthetic frame: the next power of two holding the largest axis extent, minus one
    (511 for the 512 x 512 x 256 room, 1023 for a 10-bit body)"""
    p = np.asarray(points).astype(np.int64)
    ext = int((p.max(0) - p.min(0)).max()) + 1
    return (1 << max(int(np.ceil(np.log2(max(ext, 2)))), 1)) - 1


def d1_psnr(a_points, b_points, peak):
    """MPEG pc_error D1 (point-to-point), symmetric: returns (psnr_db, mse_a_to_b, mse_b_to_a); +inf for equal sets"""
    a = np.asarray(a_points, dtype=np.float64)
    b = np.asarray(b_points, dtype=np.float64)
    e_ab = float(_nn(a, b)[0].mean())
    e_ba = float(_nn(b, a)[0].mean())
    m = max(e_ab, e_ba)
    return (float("inf") if m == 0.0 else 10.0 * np.log10(3.0 * float(peak) ** 2 / m)), e_ab, e_ba


def y_psnr(a_points, a_colors, b_points, b_colors):
    """luma PSNR over nearest-neighbour pairs, symmetric (pc_error's colour metric on Y): (psnr_db, mse_ab, mse_ba)"""
    a = np.asarray(a_points, dtype=np.float64)
    b = np.asarray(b_points, dtype=np.float64)
    ya = np.asarray(a_colors, dtype=np.float64) @ BT709
    yb = np.asarray(b_colors, dtype=np.float64) @ BT709
    c_ab = float(((ya - yb[_nn(a, b)[1]]) ** 2).mean())
    c_ba = float(((yb - ya[_nn(b, a)[1]]) ** 2).mean())
    m = max(c_ab, c_ba)
    return (float("inf") if m == 0.0 else 10.0 * np.log10(1.0 / m)), c_ab, c_ba


def frame_quality(src, rec, peak=None):
    """{"d1_psnr", "y_psnr"} of a decoded frame against its source frame ({"points", "colors"} each)"""
    peak = peak_of(src["points"]) if peak is None else peak
    return {"d1_psnr": d1_psnr(src["points"], rec["points"], peak)[0],
            "y_psnr": y_psnr(src["points"], src["colors"], rec["points"], rec["colors"])[0],
            "peak": int(peak)}
