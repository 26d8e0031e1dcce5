#!/usr/bin/env python3
"""Where a step's wall time goes on the calling thread: the library call itself against everything around it in
compress() / decompress() (argument marshalling, bytes objects, result arrays, dictionaries).
python tools/host_overhead.py [container_version]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"


def main():
    cv = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    pkg = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workloads")
    native = importlib.import_module(PKG + ".native")
    s = [[1.0, 0.0], [0.0, 1.0], [1, 1]]
    f = wl.room(1_000_000, seed=0)
    enc, dec = pkg.CompressionPipeline(s, slots=1, container_version=cv), pkg.DecompressionPipeline(slots=1)
    t_lib = {"enc": [], "dec": []}
    lib = native.NativeCodec.__dict__
    orig_e, orig_d = native.NativeCodec.encode_host_frames, native.NativeCodec.decode

    def enc_wrap(self, *a, **k):
        t = time.perf_counter()
        r = orig_e(self, *a, **k)
        t_lib["enc"].append(time.perf_counter() - t)
        return r

    def dec_wrap(self, *a, **k):
        t = time.perf_counter()
        r = orig_d(self, *a, **k)
        t_lib["dec"].append(time.perf_counter() - t)
        return r

    native.NativeCodec.encode_host_frames, native.NativeCodec.decode = enc_wrap, dec_wrap
    te, td, tg = [], [], []
    for i in range(30):
        t0 = time.perf_counter()
        out, side = enc.compress({"frames": [dict(f)], "timestamps": {}})
        t1 = time.perf_counter()
        rec, ds = dec.decompress(out[3])
        t2 = time.perf_counter()
        te.append(t1 - t0)
        td.append(t2 - t1)
    med = lambda v: 1e3 * float(np.median(v[8:]))     # noqa: E731
    nat_e = 1e3 * sum(v for k, v in side["enc_time_measurements"].items() if not isinstance(v, list))
    nat_d = 1e3 * sum(ds["time_measurements"].values())
    print(f"container version {cv}: compress() {med(te):.3f} ms, of which the native wrapper {med(t_lib['enc']):.3f} "
          f"(library's own stage clocks of the last call: {nat_e:.3f}); decompress() {med(td):.3f} ms, wrapper "
          f"{med(t_lib['dec']):.3f} (stage clocks {nat_d:.3f})")


if __name__ == "__main__":
    main()
