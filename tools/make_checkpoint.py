#!/usr/bin/env python3
"""Generate the `demo_small` checkpoint used by both the HIP path and the oracle.

The reference loads `unified/results/demo_small/{config.yaml,weights.pt}`
(sender/encoder/codec_pipeline.py:56-72), which are NOT in the reference tree
and cannot be fetched (no network).  This script therefore writes a checkpoint
of the same *shape of information* with seeded synthetic weights:

  * layer weights / biases of the architecture frozen in DESIGN.md (MODEL
    section), drawn from a counter-based integer hash (splitmix64) so the file
    is bit-reproducible on any machine;
  * the integer CDF tables that CompressAI's `model.update()` would build
    (codec_pipeline.py:69) — EntropyBottleneck and GaussianConditional — built
    here once and stored as integers, so encoder, decoder and oracle never
    recompute transcendental functions ([RECALL] CompressAI 1.2.4
    entropy_models.py / ops.cpp pmf_to_quantized_cdf).

Output: demo-learned-point-cloud-compression_amd/assets/demo_small.npz
Run:    python tools/make_checkpoint.py
"""
import os
import sys
import numpy as np
from scipy.special import erfc
from scipy.stats import norm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "demo-learned-point-cloud-compression_amd", "assets", "demo_small.npz")

C = 32      # hidden width
CY = 32     # latent channels (y)
CZ = 32     # hyper-latent channels (z)
SCALE_MIN, SCALE_MAX, SCALE_LEVELS = 0.11, 256.0, 64
TAIL_MASS = 1e-9
PRECISION = 16


# ------------------------------------------------------------------ weights
def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(name: str, shape, bound: float) -> np.ndarray:
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(_fnv1a(name))
        bits = _splitmix64(idx) >> np.uint64(40)          # 24 random bits
    u = bits.astype(np.float64) / float(1 << 24)          # exact
    return ((u - 0.5) * 2.0 * bound).astype(np.float32).reshape(shape)


def conv(name, k, cin, cout, p_eff, gain=1.0, bias_mid=0.0, bias_spread=0.05):
    bound = gain * np.sqrt(6.0 / (p_eff * cin))
    w = uniform(name + ".weight", (k, cin, cout), bound)
    b = (uniform(name + ".bias", (cout,), bias_spread) + np.float32(bias_mid)).astype(np.float32)
    return {name + ".weight": w, name + ".bias": b}


def build_weights():
    t = {}
    # g_a: (1,r,g,b) -> y, stride 1 -> 8
    t.update(conv("g_a.conv0", 27, 4, C, 9))
    t.update(conv("g_a.down0", 8, C, C, 4))
    t.update(conv("g_a.conv1", 27, C, C, 9))
    t.update(conv("g_a.down1", 8, C, C, 4))
    t.update(conv("g_a.conv2", 27, C, C, 9))
    t.update(conv("g_a.down2", 8, C, C, 4))
    t.update(conv("g_a.conv3", 27, C, CY, 9, gain=1.0))
    # g_s: y_hat -> (r,g,b), stride 8 -> 1, three generative stages
    for j in range(3):
        t.update(conv(f"g_s.up{j}", 8, CY if j == 0 else C, C, 1))
        t.update(conv(f"g_s.conv{j}", 27, C, C, 14))
        occ = conv(f"g_s.occ{j}", 1, C, 1, 1)
        t[f"g_s.occ{j}.weight"] = occ[f"g_s.occ{j}.weight"][0]
        t[f"g_s.occ{j}.bias"] = occ[f"g_s.occ{j}.bias"]
    col = conv("g_s.color", 1, C, 3, 1, bias_mid=0.5)
    t["g_s.color.weight"] = col["g_s.color.weight"][0]
    t["g_s.color.bias"] = col["g_s.color.bias"]
    # h_a: y -> z, stride 8 -> 32
    t.update(conv("h_a.conv0", 27, CY, C, 9))
    t.update(conv("h_a.down0", 8, C, C, 4))
    t.update(conv("h_a.down1", 8, C, CZ, 4, gain=0.4))
    # h_s: z_hat -> (scales | means), stride 32 -> 8
    t.update(conv("h_s.up0", 8, CZ, C, 1))
    t.update(conv("h_s.up1", 8, C, C, 1))
    hs = conv("h_s.conv0", 27, C, 2 * CY, 14, gain=0.05)
    b = hs["h_s.conv0.bias"].copy()
    b[:CY] += np.float32(1.5)                       # scales_hat centred near 1.5
    hs["h_s.conv0.bias"] = b
    t.update(hs)
    # scale_nn: q[1,2] -> [1,CY];  scale = 0.5 + |relu(q W0 + b0) W1 + b1|
    t["scale_nn.l0.weight"] = uniform("scale_nn.l0.weight", (2, 16), 1.0)
    t["scale_nn.l0.bias"] = uniform("scale_nn.l0.bias", (16,), 0.5)
    t["scale_nn.l1.weight"] = uniform("scale_nn.l1.weight", (16, CY), 0.5)
    t["scale_nn.l1.bias"] = uniform("scale_nn.l1.bias", (CY,), 0.5)
    return t


# ------------------------------------------------------------------ CDF tables
def pmf_to_quantized_cdf(pmf, precision=PRECISION):
    """[RECALL] compressai/cpp_exts/ops/ops.cpp pmf_to_quantized_cdf."""
    pmf = np.asarray(pmf, dtype=np.float32)
    assert np.all(np.isfinite(pmf)) and np.all(pmf >= 0)
    cdf = np.zeros(len(pmf) + 1, dtype=np.uint64)
    # std::round on float: half away from zero (values are >= 0)
    cdf[1:] = np.floor(pmf.astype(np.float32) * np.float32(1 << precision) + np.float32(0.5)).astype(np.uint64)
    total = int(cdf.sum())
    assert total > 0
    cdf = (np.uint64(1 << precision) * cdf) // np.uint64(total)
    cdf = np.cumsum(cdf).astype(np.int64)
    cdf[-1] = 1 << precision
    n = len(cdf)
    for i in range(n - 1):
        if cdf[i] == cdf[i + 1]:
            freq = np.diff(cdf)
            cand = np.where(freq > 1, freq, np.iinfo(np.int64).max)
            best = int(np.argmin(cand))              # first smallest freq > 1
            assert cand[best] != np.iinfo(np.int64).max
            if best < i:
                cdf[best + 1:i + 1] -= 1
            else:
                assert best > i
                cdf[i + 1:best + 1] += 1
    assert np.all(np.diff(cdf) > 0) and cdf[0] == 0 and cdf[-1] == (1 << precision)
    return cdf.astype(np.int32)


def gaussian_tables():
    """[RECALL] GaussianConditional.update()."""
    table = np.exp(np.linspace(np.log(SCALE_MIN), np.log(SCALE_MAX), SCALE_LEVELS)).astype(np.float32)
    multiplier = np.float32(-norm.ppf(TAIL_MASS / 2))
    center = np.ceil(table * multiplier).astype(np.int32)
    length = 2 * center + 1
    max_len = int(length.max())
    samples = np.abs(np.arange(max_len, dtype=np.int32)[None, :] - center[:, None]).astype(np.float32)
    scale = table[:, None].astype(np.float32)

    def phi(x):
        return (np.float32(0.5) * erfc(np.float32(-(2 ** -0.5)) * x.astype(np.float32))).astype(np.float32)

    upper = phi((np.float32(0.5) - samples) / scale)
    lower = phi((np.float32(-0.5) - samples) / scale)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdfs = np.zeros((SCALE_LEVELS, max_len + 2), dtype=np.int32)
    for i in range(SCALE_LEVELS):
        prob = np.concatenate([pmf[i, :length[i]], tail[i]])
        c = pmf_to_quantized_cdf(prob)
        cdfs[i, :len(c)] = c
    return table, cdfs, (length + 2).astype(np.int32), (-center).astype(np.int32)


def bottleneck_tables():
    """EntropyBottleneck.update() [RECALL] with an analytic (logistic) per-channel
    density standing in for the learned `_logits_cumulative`:
    logits_cumulative_c(x) = (x - m_c) / s_c."""
    m = uniform("entropy_bottleneck.median", (CZ,), 0.4).astype(np.float32)
    s = (np.float32(1.2) + np.abs(uniform("entropy_bottleneck.scale", (CZ,), 1.5))).astype(np.float32)
    target = np.float32(np.log(2.0 / TAIL_MASS - 1.0))
    q0, q2 = m - s * target, m + s * target
    medians = m
    minima = np.maximum(np.ceil(medians - q0), 0).astype(np.int32)
    maxima = np.maximum(np.ceil(q2 - medians), 0).astype(np.int32)
    pmf_start = medians - minima.astype(np.float32)
    pmf_length = maxima + minima + 1
    max_len = int(pmf_length.max())
    samples = pmf_start[:, None] + np.arange(max_len, dtype=np.float32)[None, :]

    def logits(x):
        return ((x - m[:, None]) / s[:, None]).astype(np.float32)

    def sigmoid(x):
        return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)

    lower = logits(samples - np.float32(0.5))
    upper = logits(samples + np.float32(0.5))
    sign = -np.sign(lower + upper)
    pmf = np.abs(sigmoid(sign * upper) - sigmoid(sign * lower))
    cdfs = np.zeros((CZ, max_len + 2), dtype=np.int32)
    for i in range(CZ):
        L = int(pmf_length[i])
        tail = sigmoid(lower[i, :1]) + sigmoid(-upper[i, L - 1:L])
        c = pmf_to_quantized_cdf(np.concatenate([pmf[i, :L], tail]))
        cdfs[i, :len(c)] = c
    return medians, cdfs, (pmf_length + 2).astype(np.int32), (-minima).astype(np.int32)


def main():
    t = build_weights()
    table, g_cdf, g_len, g_off = gaussian_tables()
    med, b_cdf, b_len, b_off = bottleneck_tables()
    t["gaussian_conditional.scale_table"] = table
    t["gaussian_conditional.quantized_cdf"] = g_cdf
    t["gaussian_conditional.cdf_length"] = g_len
    t["gaussian_conditional.offset"] = g_off
    t["entropy_bottleneck.medians"] = med
    t["entropy_bottleneck.quantized_cdf"] = b_cdf
    t["entropy_bottleneck.cdf_length"] = b_len
    t["entropy_bottleneck.offset"] = b_off
    t["entropy_model.eps"] = np.float32(1e-3)
    t["entropy_model.offsets_ab"] = np.array([0.15, 0.3], dtype=np.float32)   # get_offsets = a / (b + sigma)
    t["config.channels"] = np.array([C, CY, CZ], dtype=np.int32)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **t)
    nbytes = os.path.getsize(OUT)
    print(f"wrote {OUT}: {len(t)} arrays, {nbytes/1e6:.2f} MB")


if __name__ == "__main__":
    sys.exit(main())
