#!/usr/bin/env python3
"""Every GPU activity of one of the last steps of a trace collected as tools/timeline.py says: start, duration, gap to the
previous activity's end.  tools/timeline_dump.py <dir> [steps_back] [from_us] [to_us]"""
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
import timeline  # noqa: E402

ev = timeline.load(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi_us = float(sys.argv[4]) if len(sys.argv) > 4 else 1e12
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_frames_keys")]
seg = ev[starts[-back - 1]:starts[-back]]
t0 = seg[0][0]
prev_e = t0
for s, e, n in seg:
    if lo_us <= (s - t0) / 1e3 <= hi_us:
        print(f"+{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_e) / 1e3:7.1f}  {n[:70]}")
    prev_e = max(prev_e, e)
