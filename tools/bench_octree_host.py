#!/usr/bin/env python3
"""Host halves of geometry blob version 1 (csrc/octree_host.cpp) on the latent of the 1M-point bench frame (26k leaves,
8.7k nodes, 65k binary decisions): pcc_octree_pack from the oracle's occupancy bytes, pcc_octree_unpack of the blob.
No GPU needed: python tools/bench_octree_host.py"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"


def main():
    abi = importlib.import_module(PKG + "._abi")
    wl = importlib.import_module(PKG + ".workloads")
    from oracle.codec_ref import Oracle
    lib = abi.lib()
    o = Oracle(threads=4)
    room = wl.room(1_000_000, seed=0)
    lat = np.unique(room["points"].astype(np.int64) // 8, axis=0).astype(np.int32)
    blob = o.octree_encode(lat, 4096, version=1)
    buf = np.frombuffer(blob, np.uint8)
    n = lat.shape[0]
    pts = np.empty((n, 3), np.int32)
    level_n = (C.c_int64 * 16)()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = lib.pcc_octree_unpack_levels(buf.ctypes.data, len(blob), pts.ctypes.data, n, level_n)
    t_dec = (time.perf_counter() - t0) / reps
    assert rc == 0 and np.array_equal(pts, o.octree_decode(blob))
    depth = blob[2]
    # occupancy bytes for the pack: decode them with a tiny python walk of the oracle's points (breadth-first)
    keys = np.zeros(n, np.uint64)
    q = (pts - np.array([int.from_bytes(blob[8 + 4 * a:12 + 4 * a], "little", signed=True) for a in range(3)])).astype(np.uint64)
    for b in range(depth):
        for a in range(3):
            keys |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + (2 - a))
    keys = np.sort(keys)
    occ, ln = [], []
    for L in range(depth):
        node = keys >> np.uint64(3 * (depth - L))
        child = ((keys >> np.uint64(3 * (depth - L - 1))) & np.uint64(7)).astype(np.int64)
        un, inv = np.unique(node, return_inverse=True)
        byte = np.zeros(un.shape[0], np.int64)
        np.bitwise_or.at(byte, inv, 1 << child)
        occ.append(byte.astype(np.uint8))
        ln.append(len(byte))
    occ = np.concatenate(occ)
    lvl = (C.c_int64 * depth)(*ln)
    org = (C.c_int * 3)(*[int.from_bytes(blob[8 + 4 * a:12 + 4 * a], "little", signed=True) for a in range(3)])
    out = np.empty(len(occ) * 2 + 128, np.uint8)
    ol = C.c_int64(0)
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = lib.pcc_octree_pack(occ.ctypes.data, lvl, depth, n, org, out.ctypes.data, out.shape[0], C.byref(ol))
    t_enc = (time.perf_counter() - t0) / reps
    assert rc == 0 and out[:ol.value].tobytes() == blob
    print(f"latent of the 1M-point frame: {n} leaves, {len(occ)} nodes, blob {len(blob)} B: pack {1e3 * t_enc:.3f} ms, unpack {1e3 * t_dec:.3f} ms")


if __name__ == "__main__":
    main()
