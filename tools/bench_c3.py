#!/usr/bin/env python3
"""C3 (BASELINE.json configs[2]): geometry-only octree occupancy coding of a KITTI-like sweep on one MI355X.
Times utils.gpcc_encode / gpcc_decode (the slot of shared/utils.py:169-240) from Morton-sorted keys in HBM to the
blob on the host and back to host coordinates, with the device / host split of the version-1 blob."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"


def main():
    pkg = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workloads")
    utils = importlib.import_module(PKG + ".utils")
    rtm = importlib.import_module(PKG + "._abi") and importlib.import_module(PKG + ".runtime")
    rt = rtm.Runtime(0)
    reps = int(os.environ.get("REPS", "10"))
    out = {}
    with rt:
        for name, frame in (("C3 lidar sweep", wl.lidar_sweep()), ("room 1M stride 1", wl.room(1_000_000, seed=0))):
            pts = frame["points"].astype(np.int32)
            coords = np.concatenate([np.zeros((pts.shape[0], 1), np.int32), pts], 1)
            keys = rt.morton_keys(rt.to_device(coords))
            rt.sort_pairs(keys)
            kh = keys.cpu().numpy()
            n = kh.shape[0]
            first, last = int(kh[0]) & 0xFFFFFFFFFFFFFFFF, int(kh[-1]) & 0xFFFFFFFFFFFFFFFF
            depth, origin = utils.octree_depth_origin(first, last, 0)
            res = {"points": n, "depth": depth}
            for version in (1, 2):
                t_enc, t_dec = [], []
                for it in range(reps + 2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    blob = rt.octree_encode(keys, 0, version=version)
                    t1 = time.perf_counter()
                    dec = rt.octree_decode(blob)
                    t2 = time.perf_counter()
                    assert dec.shape[0] == n
                    if it >= 2:
                        t_enc.append(t1 - t0); t_dec.append(t2 - t1)
                med = lambda v: 1e3 * float(np.median(v))     # noqa: E731
                res[f"v{version}"] = {"blob_bytes": len(blob), "bpp": 8 * len(blob) / n, "encode_ms": med(t_enc),
                                      "decode_ms": med(t_dec)}
            # the operator pair (utils.gpcc_encode / gpcc_decode: the version rule picks 2 above 65536 leaves)
            t_enc, t_dec = [], []
            for it in range(reps + 2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                blob = utils.gpcc_encode(keys, kh, 0, n, 0)
                t1 = time.perf_counter()
                dec = utils.gpcc_decode(blob, 1)
                t2 = time.perf_counter()
                if it >= 2:
                    t_enc.append(t1 - t0); t_dec.append(t2 - t1)
            res["operator"] = {"blob_version": blob[1], "encode_ms": med(t_enc), "decode_ms": med(t_dec)}
            rt.prof_enable(True, reserve=64)
            blob = rt.octree_encode(keys, 0, version=2)
            rt.octree_decode(blob)
            torch.cuda.synchronize()
            res["device_ms"] = {op: round(ms, 4) for op, ms, _ in rt.prof_records()}
            rt.prof_enable(False)
            out[name] = res
            print(name, json.dumps(res), flush=True)
    rt.close()


if __name__ == "__main__":
    main()
