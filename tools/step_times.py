#!/usr/bin/env python3
"""Per-step wall times of compress / decompress on the 1M-point room (host numpy in / out), one line per step:
which steps are the slow ones.  python tools/step_times.py [steps] [container_version]"""
import gc
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"


def throttled(tag):
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        print(f"[cgroup] {tag}: usage {int(d['usage_usec']) / 1e6:.2f} s, throttled {d['nr_throttled']} times, "
              f"{int(d['throttled_usec']) / 1e3:.0f} ms", flush=True)
    except Exception:
        pass


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cv = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    throttled("start")
    pkg = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workloads")
    throttled("imports")
    s = [[1.0, 0.0], [0.0, 1.0], [1, 1]]
    f = wl.room(1_000_000, seed=0)
    throttled("workload")
    enc, dec = pkg.CompressionPipeline(s, slots=1, container_version=cv), pkg.DecompressionPipeline(slots=1)
    throttled("pipelines")
    rows = []
    if os.environ.get("STEP_TIMES_GC") == "0":     # as bench.py's timed regions: collected, then disabled
        for _ in range(6):
            out, _ = enc.compress({"frames": [dict(f)], "timestamps": {}})
            dec.decompress(out[3])
        gc.collect()
        gc.disable()
    throttled("warm")
    for i in range(n):
        g0 = gc.get_count()
        t0 = time.perf_counter()
        out, _ = enc.compress({"frames": [dict(f)], "timestamps": {}})
        t1 = time.perf_counter()
        rec, side = dec.decompress(out[3])
        t2 = time.perf_counter()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), g0, {k: round(1e3 * v, 2) for k, v in side["time_measurements"].items()}))
    throttled("steps done")
    for i, r in enumerate(rows):
        flag = " <--" if i > 3 and (r[0] > 1.3 * sorted(x[0] for x in rows)[n // 2] or r[1] > 1.3 * sorted(x[1] for x in rows)[n // 2]) else ""
        print(f"step {i:3d}: encode {r[0]:7.3f} ms  decode {r[1]:7.3f} ms  gc {r[2]}{flag}  {r[3] if flag else ''}")


if __name__ == "__main__":
    main()
