"""GPU voxelisation of a camera frame (capturer.py:88-126) against the numpy
restatement in oracle/capture_ref.py."""
import numpy as np
import pytest

from conftest import pkg


def camera_frame(seed, w=160, h=120, far=False):
    """ZED-like XYZRGBA float32 [w*h,4]: a wall, a sphere and a floor seen from the origin, with NaN /
    inf pixels and some points beyond the depth clip"""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.linspace(-0.6, 0.6, w), np.linspace(-0.45, 0.45, h))
    d = np.stack([u, v, -np.ones_like(u)], -1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = np.full(d.shape[0], 1.2)                                    # back wall at z = -1.2
    t = np.where(d[:, 1] < -0.05, np.minimum(t, -0.35 / np.minimum(d[:, 1], -1e-6)), t)   # floor y = -0.35
    c = np.array([0.1, 0.0, -0.8])
    b = d @ c
    disc = b * b - (c @ c - 0.2 ** 2)
    ts = np.where(disc > 0, b - np.sqrt(np.maximum(disc, 0)), np.inf)
    t = np.where((ts > 0) & (ts < t), ts, t)
    p = (d * t[:, None] + rng.normal(0, 0.0008, d.shape)).astype(np.float32)
    if far:
        p[::17] *= 3.0                                              # beyond depth_clip
    p[::101] = np.nan
    p[5::211, 0] = np.inf
    rgba = (rng.integers(0, 256, d.shape[0], dtype=np.uint32) | (rng.integers(0, 256, d.shape[0], dtype=np.uint32) << 8)
            | (rng.integers(0, 256, d.shape[0], dtype=np.uint32) << 16) | (np.uint32(255) << 24))
    return np.concatenate([p, rgba.view(np.float32)[:, None]], 1).astype(np.float32)


def test_oracle_matches_dictionary_form():
    """the sorted-segment restatement and the one-accumulator-per-voxel form agree"""
    from oracle import capture_ref as ref
    data = camera_frame(1, 80, 60, far=True)
    out = ref.voxelize(data, 1.4, 0.01)
    d = ref.voxelize_open3d_semantics(data, 1.4, 0.01)
    pts = [tuple(int(v) for v in p) for p in out["points"]]
    assert sorted(pts) == sorted(d.keys()) and pts == sorted(pts)
    for p, col in zip(pts, out["colors"]):
        assert any(np.array_equal(np.asarray(c), col) for c in d[p])
    assert out["colors"].min() >= 0 and out["colors"].max() <= 1
    capped = ref.voxelize(data, 1.4, 0.01, max_points=500)
    assert capped["points"].shape[0] == 500
    assert capped["points"][:, 2].min() >= np.sort(out["points"][:, 2])[-500]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,voxel,max_points,far", [(1, 0.005, None, True), (2, 0.01, 3000, False),
                                                        (3, 0.02, None, True), (4, 0.005, 30000, True)])
def test_voxelize_gpu_equals_oracle(rt, seed, voxel, max_points, far):
    from oracle import capture_ref as ref
    capture = pkg("capture")
    data = camera_frame(seed, far=far)
    out = capture.voxelize(rt, data, 1.4, voxel, max_points)
    exp = ref.voxelize(data, 1.4, voxel, max_points)
    assert out["points"].dtype == np.int16 and out["colors"].dtype == np.float64
    assert np.array_equal(out["points"], exp["points"])
    assert np.array_equal(out["colors"], exp["colors"])
    assert np.unique(out["points"], axis=0).shape[0] == out["points"].shape[0]


@pytest.mark.gpu
def test_voxelize_edge_cases(rt):
    capture = pkg("capture")
    assert capture.voxelize(rt, np.zeros((0, 4), np.float32))["points"].shape == (0, 3)
    allnan = np.full((50, 4), np.nan, np.float32)
    assert capture.voxelize(rt, allnan)["points"].shape == (0, 3)
    one = np.array([[0.1, -0.2, -0.5, 0.0]], np.float32)
    one[0, 3] = np.array([0x00FF8040], np.uint32).view(np.float32)[0]
    out = capture.voxelize(rt, one, 1.4, 0.005)
    assert out["points"].tolist() == [[20, -40, -100]]
    assert np.allclose(out["colors"], [[0x40 / 255, 0x80 / 255, 0xFF / 255]], rtol=0, atol=0)


@pytest.mark.gpu
def test_voxelized_frame_goes_through_the_codec(rt):
    """capture pre-step -> compress -> decompress, the chain of the demo's sender and receiver"""
    capture = pkg("capture")
    wl = pkg("workloads")
    frame = capture.voxelize(rt, camera_frame(7), 1.4, 0.005, 30000)
    enc = pkg("codec_pipeline").CompressionPipeline([[1, 1]], slots=1)
    dec = pkg("codec_parallel").DecompressionPipeline(slots=1)
    out, side = enc.compress(wl.gop([frame]))
    rec, _ = dec.decompress(out[1])
    assert rec[0]["points"].shape[0] == frame["points"].shape[0]
