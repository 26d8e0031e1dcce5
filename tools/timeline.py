#!/usr/bin/env python3
"""Where the GPU idles inside a codec step.  Collect with
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/count_launches.py
then `python tools/timeline.py <dir> [steps_back]`: the busy time, the span and the largest gaps of one of the last
steps (a step = from the encoder's key kernel to the last activity before the next one)."""
import csv
import glob
import sys


def load(src):
    ev = []
    for fn in glob.glob(f"{src}/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    for fn in glob.glob(f"{src}/**/*_memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?")))
    ev.sort()
    return ev


def main():
    src = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ev = load(src)
    # the encoder's and the decoder's key kernels alternate; an encode starts with the frame-key kernel
    starts = [i for i, e in enumerate(ev) if e[2].startswith("k_frames_keys")]
    lo, hi = starts[-back - 1], starts[-back]
    seg = ev[lo:hi]
    t0 = seg[0][0]
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    gaps = []
    for s, e, name in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = cur_e - t0
    print(f"step: {len(seg)} GPU activities, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, "
          f"idle {(span - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
    prev = {g[1]: None for g in gaps}
    names = {}
    for i, (s, e, name) in enumerate(seg):
        names[e - t0] = name
    for g, at, nxt in sorted(gaps, reverse=True)[:18]:
        print(f"  gap {g / 1e3:8.1f} us at +{at / 1e6:.3f} ms   after {names.get(at, '?'):40s} before {nxt}")
    small = sum(g for g, _, _ in gaps if g < 20000)
    print(f"  gaps under 20 us: {sum(1 for g in gaps if g[0] < 20000)} totalling {small / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
