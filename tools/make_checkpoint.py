#!/usr/bin/env python3
"""Generate the `demo_small` checkpoint used by both the HIP path and the oracle.

The reference loads `unified/results/demo_small/{config.yaml,weights.pt}`
(sender/encoder/codec_pipeline.py:56-72), which are NOT in the reference tree
and cannot be fetched (no network).  This script therefore writes a checkpoint
of the same *shape of information* with seeded synthetic weights:

  * layer weights / biases of the architecture frozen in DESIGN.md (MODEL
    section), drawn from a counter-based integer hash (splitmix64) so the file
    is bit-reproducible on any machine;
  * the raw parameters of the two entropy models in CompressAI's state_dict form
    (EntropyBottleneck matrices / biases / quantiles, GaussianConditional scale
    table) and the integer CDF tables the product's `model.update()` builds from
    them (codec_pipeline.py:69; demo-learned-point-cloud-compression_amd/tables.py)
    — stored too, so that loading the in-tree asset evaluates no transcendental
    function; `--model-dir` writes a model directory with the raw parameters only.

Output: demo-learned-point-cloud-compression_amd/assets/demo_small.npz
Run:    python tools/make_checkpoint.py
"""
import argparse
import importlib
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "demo-learned-point-cloud-compression_amd", "assets", "demo_small.npz")
tables = importlib.import_module("demo-learned-point-cloud-compression_amd.tables")


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


def build_weights(C=32, CY=32, CZ=32):
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


# ------------------------------------------------------------------ entropy parameters
def entropy_raw(cz):
    """Raw parameters of the two entropy models in CompressAI's state_dict form; the integer CDF tables are built
    from them by the product's model.update() (demo-learned-point-cloud-compression_amd/tables.py).
    EntropyBottleneck: an analytic per-channel logistic density stands in for a learned one — filters = () (one
    layer): logits_cumulative_c(x) = softplus(_matrix0_c) x + _bias0_c with softplus(_matrix0_c) = 1 / s_c and
    _bias0_c = -m_c / s_c; quantiles at tail mass 1e-9."""
    m = uniform("entropy_bottleneck.median", (cz,), 0.4).astype(np.float32)
    s = (np.float32(1.2) + np.abs(uniform("entropy_bottleneck.scale", (cz,), 1.5))).astype(np.float32)
    target = np.float32(np.log(2.0 / tables.TAIL_MASS - 1.0))
    inv = (np.float32(1) / s).astype(np.float32)
    t = {"entropy_bottleneck._matrix0": np.log(np.expm1(inv)).astype(np.float32).reshape(cz, 1, 1),
         "entropy_bottleneck._bias0": (-m * inv).astype(np.float32).reshape(cz, 1, 1),
         "entropy_bottleneck.quantiles": np.stack([m - s * target, m, m + s * target], 1).astype(np.float32).reshape(cz, 1, 3),
         "gaussian_conditional.scale_table": tables.default_scale_table()}
    return t


def build(c=32, cy=32, cz=32, with_tables=True):
    t = build_weights(c, cy, cz)
    t.update(entropy_raw(cz))
    if with_tables:
        tables.update_tensors(t)        # what model.update() does at load time
    t["entropy_model.eps"] = np.float32(1e-3)
    t["entropy_model.offsets_ab"] = np.array([0.15, 0.3], dtype=np.float32)   # get_offsets = a / (b + sigma)
    t["config.channels"] = np.array([c, cy, cz], dtype=np.int32)
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=32, help="hidden width C")
    ap.add_argument("--latent", type=int, default=32, help="latent channels C_y")
    ap.add_argument("--hyper", type=int, default=32, help="hyper-latent channels C_z")
    ap.add_argument("--model-dir", default=None,
                    help="write <model-dir>/{config.yaml,weights.npz} (what load_model reads: raw entropy parameters, "
                         "no integer tables) instead of the in-tree asset")
    args = ap.parse_args()
    if args.model_dir:
        write_model_dir(args.model_dir, args.channels, args.latent, args.hyper)
        return 0
    t = build(args.channels, args.latent, args.hyper)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **t)
    nbytes = os.path.getsize(OUT)
    print(f"wrote {OUT}: {len(t)} arrays, {nbytes/1e6:.2f} MB")
    return 0


def write_model_dir(path, c, cy, cz, name=None):
    """a model directory as the reference's load_model expects it (codec_pipeline.py:56-72): config.yaml with the
    `model` section + the weights — here weights.npz with RAW entropy parameters: update() builds the tables"""
    import yaml
    os.makedirs(path, exist_ok=True)
    t = build(c, cy, cz, with_tables=False)
    np.savez_compressed(os.path.join(path, "weights.npz"), **t)
    cfg = {"model": {"name": name or os.path.basename(os.path.normpath(path)), "channels": int(c),
                     "latent_channels": int(cy), "hyper_channels": int(cz)}}
    with open(os.path.join(path, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    return t


if __name__ == "__main__":
    sys.exit(main())
