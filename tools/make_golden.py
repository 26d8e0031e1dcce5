#!/usr/bin/env python3
"""Generate tests/golden/*.npz: inputs + expected outputs of the codec path.

Expected outputs come from the CPU oracle (oracle/codec_ref.py) run in the
build container; the reference itself cannot run here, so these
vectors pin the oracle and the HIP path to each other and across rounds — they
are NOT reference outputs.  Inputs: the C1 synthetic sphere and two recorded
ZED frames from the reference's evaluation data (data files only; colours
stored as the uint8 they were captured as: colour = k/255); zed_seq25: the 25
frames the reference's encoder service samples from the first five seconds of
evaluation/data/test_sequence (5 frames per 1-s segment, shared/config.yaml:9-15),
with digests of the oracle's containers and reconstructions per GOP.

Run: python tools/make_golden.py        (needs /root/reference for the ZED case)
"""
import hashlib
import importlib
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.codec_ref import Oracle  # noqa: E402

wl = importlib.import_module("demo-learned-point-cloud-compression_amd.workloads")
OUT = os.path.join(ROOT, "tests", "golden")
SETTINGS = [[1.0, 0.0], [0.0, 1.0], [1, 1]]          # shared/config.yaml:12-15


def digest(frames):
    h = hashlib.sha256()
    for f in frames:
        h.update(np.ascontiguousarray(f["points"], dtype=np.int32).tobytes())
        h.update(np.ascontiguousarray(f["colors"], dtype=np.float32).tobytes())
    return h.hexdigest()


def emit(name, frames, oracle):
    out, dbg = oracle.compress(frames, SETTINGS)
    rec = {"n_frames": np.int32(len(frames)), "settings": np.asarray(SETTINGS, dtype=np.float64)}
    for i, f in enumerate(frames):
        rec[f"points_{i}"] = np.asarray(f["points"], dtype=np.int16)
        rec[f"colors_u8_{i}"] = np.rint(np.asarray(f["colors"]) * 255).astype(np.uint8)
    for q, b in out.items():
        rec[f"container_{q}"] = np.frombuffer(b, dtype=np.uint8)
        rec[f"decoded_sha256_{q}"] = np.frombuffer(digest(oracle.decompress(b)).encode(), dtype=np.uint8)
    rec["k"] = np.asarray(dbg["k"], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "points", [f["points"].shape[0] for f in frames], "bytes", {q: len(b) for q, b in out.items()})


def zed_frames(paths):
    frames = []
    for p in paths:
        with open(p, "rb") as fh:
            d = pickle.load(fh)
        col = np.rint(np.asarray(d["colors"]) * 255).astype(np.uint8).astype(np.float64) / 255.0
        frames.append({"points": np.asarray(d["points"], dtype=np.int16), "colors": col})
    return frames


SEQ_GOPS = [(0, 5), (5, 10), (10, 15), (15, 20), (20, 25),             # the reference's operating point: 5 GOPs of 5
            (0, 1), (1, 4), (4, 9), (9, 11), (11, 15), (15, 20), (20, 25)]  # one pipeline pair, GOP size changing: 1 3 5 2 4 5 5


def sampled_sequence(base, seconds=5, target_fps=5, segment_duration=1.0):
    """the frames Encoder.sample() (sender/encoder/encoder.py:95-129) picks from the first `seconds` one-second segments
    of a recorded sequence: per segment target_fps frames, each the recorded frame nearest to start + i / target_fps"""
    recs = []
    for name in sorted(os.listdir(base)):
        with open(os.path.join(base, name), "rb") as fh:
            d = pickle.load(fh)
        recs.append(d)
        if d["timestamp"] - recs[0]["timestamp"] > seconds * segment_duration + 1.0:
            break
    t0 = recs[0]["timestamp"]
    out = []
    for s in range(seconds):
        seg = [d for d in recs if s * segment_duration <= d["timestamp"] - t0 < (s + 1) * segment_duration]
        n = int(segment_duration * target_fps)
        for i in range(n):
            t = seg[0]["timestamp"] + i * segment_duration / n
            out.append(min(seg, key=lambda d: abs(d["timestamp"] - t)))
    frames = []
    for d in out:
        col = np.rint(np.asarray(d["colors"]) * 255).astype(np.uint8).astype(np.float64) / 255.0
        frames.append({"points": np.asarray(d["points"], dtype=np.int16), "colors": col})
    return frames


def emit_sequence(name, frames, oracle):
    """inputs (int16 points, uint8 colours) + per GOP of SEQ_GOPS the sha256 of the three containers (both container
    versions) and of the three reconstructions"""
    rec = {"n_frames": np.int32(len(frames)), "settings": np.asarray(SETTINGS, dtype=np.float64),
           "gops": np.asarray(SEQ_GOPS, dtype=np.int32)}
    for i, f in enumerate(frames):
        rec[f"points_{i}"] = np.asarray(f["points"], dtype=np.int16)
        rec[f"colors_u8_{i}"] = np.rint(np.asarray(f["colors"]) * 255).astype(np.uint8)
    sha = lambda b: np.frombuffer(hashlib.sha256(b).hexdigest().encode(), dtype=np.uint8)   # noqa: E731
    for g, (lo, hi) in enumerate(SEQ_GOPS):
        dec0 = None
        for version in (0, 1):
            out, _ = oracle.compress(frames[lo:hi], SETTINGS, version=version)
            for q, b in out.items():
                rec[f"g{g}_v{version}_container_{q}"] = sha(b)
                rec[f"g{g}_v{version}_bytes_{q}"] = np.int64(len(b))
            dec = {q: digest(oracle.decompress(out[q])) for q in out}
            if version == 0:
                dec0 = dec
                for q in out:
                    rec[f"g{g}_decoded_{q}"] = np.frombuffer(dec[q].encode(), dtype=np.uint8)
            else:
                assert dec == dec0, "the two container versions must decode to the same frames"
        print(name, "gop", g, (lo, hi), "points", sum(f["points"].shape[0] for f in frames[lo:hi]),
              "bytes", [int(rec[f"g{g}_v0_bytes_{q}"]) for q in (1, 2, 3)], flush=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)


def octree_sets():
    """point sets of the geometry-slot vectors (lattice units, bias 4096 = a stride-8 latent): a wavy sheet at three sizes
    (below the rule of blob version 3, inside it, at its upper end), a dense block, a single point"""
    def sheet(n, seed, side):
        rng = np.random.default_rng(seed)
        out = np.zeros((0, 3), np.int32)
        while out.shape[0] < n:
            x, y = rng.integers(0, side, 2 * n), rng.integers(0, side, 2 * n)
            z = (side / 4 * (1 + np.sin(x / 17.0) * np.cos(y / 23.0))).astype(np.int64) + rng.integers(0, 3, 2 * n)
            out = np.unique(np.concatenate([out, np.stack([x, y, z], 1).astype(np.int32) - side // 3]), axis=0)
        return out[rng.permutation(out.shape[0])[:n]]
    g = np.stack(np.meshgrid(*[np.arange(12)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.int32) - 5
    return {"sheet_700": sheet(700, 1, 60), "sheet_9000": sheet(9000, 2, 160), "sheet_40000": sheet(40000, 3, 300),
            "block_1728": g, "one": np.array([[3, -2, 7]], np.int32)}


def emit_octree(oracle):
    """tests/golden/octree_blobs.npz: per set the points and the blob of every version that takes it (1, 2, 3 — version 3
    from 2 leaves) plus the version the rule picks"""
    rec = {}
    for name, pts in octree_sets().items():
        rec[f"{name}_points"] = pts
        for v in (1, 2, 3):
            if v == 3 and pts.shape[0] < 2:
                continue
            rec[f"{name}_v{v}"] = np.frombuffer(oracle.octree_encode(pts, 4096, version=v), dtype=np.uint8)
        rule = oracle.octree_encode(pts, 4096)
        rec[f"{name}_rule"] = np.int32(rule[1])
        print("octree", name, pts.shape[0], "leaves; bytes",
              {v: int(rec[f"{name}_v{v}"].shape[0]) for v in (1, 2, 3) if f"{name}_v{v}" in rec}, "rule", rule[1])
    np.savez_compressed(os.path.join(OUT, "octree_blobs.npz"), **rec)


def main():
    os.makedirs(OUT, exist_ok=True)
    o = Oracle()
    emit_octree(o)
    if "--octree-only" in sys.argv:
        return
    f = wl.sphere_shell()
    f["colors"] = np.rint(f["colors"] * 255).astype(np.uint8).astype(np.float64) / 255.0
    emit("c1_sphere", [f], o)
    base = "/root/reference/evaluation/data/test"
    if os.path.isdir(base):
        emit("zed_gop2", zed_frames([os.path.join(base, "frame_00000.pkl"), os.path.join(base, "frame_00010.pkl")]), o)
    else:
        print("reference data not present; zed_gop2 not regenerated")
    seq = "/root/reference/evaluation/data/test_sequence"
    if os.path.isdir(seq):
        emit_sequence("zed_seq25", sampled_sequence(seq), o)
    else:
        print("reference data not present; zed_seq25 not regenerated")


if __name__ == "__main__":
    main()
