"""Blob version 2 of the geometry slot (csrc/octree2.hip; oracle/pcc_oracle.c orc_octree2_encode): the occupancy
entropy coder that runs on the GPU.  CPU part: the oracle against a second, structurally different implementation
written here from the format's description and against a hand-derived stream; GPU part: the HIP coder against the
oracle, bit for bit, both directions, plus corrupt blobs."""
import ctypes as C
import struct

import numpy as np
import pytest

from conftest import pkg, random_cloud

LANES, SMAX, CTX = 64, 512, 108


# ------------------------------------------------------------------ a second implementation (pure Python)
def _occupancy_bytes(pts):
    """breadth-first occupancy bytes of distinct points int [n,3] (>= 0 after the caller's shift), x the top bit of
    a Morton triple; returns (bytes per level, depth, origin)"""
    pts = np.asarray(pts, dtype=np.int64)
    lo = pts.min(0)
    depth = 1
    while True:                                   # smallest aligned cube of the (biased) lattice holding every point
        org = (pts[0] >> depth) << depth
        if np.all((pts >> depth) == (pts[0] >> depth)):
            break
        depth += 1
    rel = pts - org
    levels = []
    for L in range(depth):
        sh = depth - L                            # node = rel >> sh, child octant = bit sh-1 of each axis
        node = rel >> sh
        octant = (((rel[:, 0] >> (sh - 1)) & 1) << 2) | (((rel[:, 1] >> (sh - 1)) & 1) << 1) | ((rel[:, 2] >> (sh - 1)) & 1)
        key = 0
        for b in range(16 - 1, -1, -1):           # Morton order of the nodes
            key = (key << 3) | (((node[:, 0] >> b) & 1) << 2) | (((node[:, 1] >> b) & 1) << 1) | ((node[:, 2] >> b) & 1)
        d = {}
        for k, o in zip(key.tolist(), octant.tolist()):
            d[k] = d.get(k, 0) | (1 << o)
        levels.append([d[k] for k in sorted(d)])
    del lo
    return levels, depth, org


def py_octree2_encode(points, bias):
    """blob version 2, written from the description in csrc/octree2.hip's header: per-step lists instead of the
    oracle's [step][lane] record table, Python integers throughout"""
    pts = np.unique(np.asarray(points, dtype=np.int64) + bias, axis=0)
    n = pts.shape[0]
    levels, depth, org = _occupancy_bytes(pts)
    occ = [b for lv in levels for b in lv]
    n_nodes = len(occ)
    nc = max(1, -(-n_nodes // (LANES * SMAX)))
    S = max(4, -(-(-(-n_nodes // (LANES * nc))) // 4) * 4)
    start_last = n_nodes - len(levels[-1])
    start_prev = start_last - len(levels[-2]) if depth >= 2 else 0

    def cls(i):
        return 0 if i >= start_last else (1 if i >= start_prev else 2)

    def decisions(i):                             # (context, bit) of node i, in coding order
        ones = 0
        for j in range(8):
            bit = (occ[i] >> j) & 1
            if j == 7 and ones == 0:
                return
            yield j, cls(i) * 36 + j * (j + 1) // 2 + ones, bit
            ones += bit

    c0, c1 = [0] * CTX, [0] * CTX
    for i in range(n_nodes):
        for _, ctx, bit in decisions(i):
            (c1 if bit else c0)[ctx] += 1
    p0 = [min(4080, max(16, (4096 * (2 * c1[k] + 1)) // (2 * (c0[k] + c1[k] + 1)))) for k in range(CTX)]
    chunks = []
    for c in range(nc):
        # forward: per lane, the (step, p1, bit) triples of its run of nodes
        per_step = {}
        for l in range(LANES):
            model = list(p0)
            for s in range(S):
                i = (LANES * c + l) * S + s
                if i >= n_nodes:
                    break
                for j, ctx, bit in decisions(i):
                    per_step.setdefault(8 * s + j, []).append((l, model[ctx], bit))
                    model[ctx] = model[ctx] + ((4096 - model[ctx]) >> 4) if bit else model[ctx] - (model[ctx] >> 4)
        x = [1 << 16] * LANES
        words = [[] for _ in range(LANES)]        # per lane, in emission order (reversed below)
        for t in sorted(per_step, reverse=True):
            for l, p1, bit in per_step[t]:
                freq, start = (p1, 4096 - p1) if bit else (4096 - p1, 0)
                if x[l] >= freq << 20:
                    words[l].append(x[l] & 0xFFFF)
                    x[l] >>= 16
                x[l] = ((x[l] // freq) << 12) + x[l] % freq + start
        states = []
        for l in range(LANES):
            states += [x[l] & 0xFFFF, x[l] >> 16]
        chunks.append(states + [len(w) for w in words] + [v for w in words for v in w[::-1]])
    body = b"".join(struct.pack("<I", len(lv)) for lv in levels) + struct.pack("<II", S, nc)
    body += struct.pack("<%dH" % CTX, *p0) + b"".join(struct.pack("<I", len(ch)) for ch in chunks)
    body += b"".join(struct.pack("<%dH" % len(ch), *ch) for ch in chunks)
    head = bytes([ord("O"), 2, depth, 0]) + struct.pack("<I", n) + struct.pack("<3i", *[int(v) - bias for v in org])
    return head + struct.pack("<I", len(body)) + body


# ------------------------------------------------------------------ CPU: oracle vs second implementation, KAT
def _clouds():
    rng = np.random.default_rng(5)
    yield "one point", np.array([[3, -4, 5]])
    yield "two neighbours", np.array([[3, -4, 5], [3, -4, 6]])
    yield "full cube 2x2x2", np.stack(np.meshgrid(*[np.arange(2)] * 3, indexing="ij"), -1).reshape(-1, 3) - 1
    yield "negative scatter", random_cloud(rng, 700, extent=90, lo=-60)[:, 1:]
    yield "dense block", np.stack(np.meshgrid(*[np.arange(12)] * 3, indexing="ij"), -1).reshape(-1, 3) + 100
    yield "far apart", np.array([[-32768, -32768, -32768], [32767, 32767, 32767], [0, 0, 0]])


@pytest.mark.parametrize("name", [n for n, _ in _clouds()])
def test_oracle_v2_equals_second_implementation(oracle, name):
    pts = dict(_clouds())[name].astype(np.int32)
    ref = oracle.octree_encode(pts, 32768, version=2)
    assert ref == py_octree2_encode(pts, 32768)
    dec = oracle.octree_decode(ref)
    assert np.array_equal(dec, oracle.octree_decode(oracle.octree_encode(pts, 32768, version=1)))
    assert np.array_equal(np.unique(dec, axis=0), np.unique(pts, axis=0)) and len(dec) == len(np.unique(pts, axis=0))


def test_v2_known_answer_one_point(oracle):
    """Worked by hand.  One point (3,-4,5): biased by 32768 every axis has bits ...; the root cube is the cell pair
    holding it (depth 1), its corner has the low bit cleared -> origin (2,-4,4) - no: (3,-4,5) -> biased (32771, 32764,
    32773): low bits (1, 0, 1) -> octant = 1*4 + 0*2 + 1 = 5, origin = (2, -4, 4).  One node, byte 1 << 5 = 0x20.
    Level class 0 (the last level), decisions j = 0..5: zeros at contexts j(j+1)/2 (ones = 0): 0,1,3,6,10, then the one
    at j = 5 (context 15), then zeros at j = 6, 7 with ones = 1: contexts 22, 29.
    p0 = (4096 (2 c1 + 1)) // (2 (c0 + c1 + 1)): a context that saw one zero -> 4096 // 4 = 1024; the one that saw one
    one -> 4096*3 // 4 = 3072; unused -> 2048.
    rANS, x0 = 65536, decisions in reverse order (lane 0 only; S = 4, one chunk):
      ctx 29 bit 0: freq 3072, start 0:    x = (65536 // 3072 << 12) + 65536 % 3072       = 21*4096 + 1024 = 87040
      ctx 22 bit 0: freq 3072:             x = (87040 // 3072 << 12) + 87040 % 3072       = 28*4096 + 1024 = 115712
      ctx 15 bit 1: freq 3072, start 1024: x = (115712 // 3072 << 12) + 115712 % 3072 + 1024 = 37*4096 + 2048 + 1024 = 154624
      ctx 10, 6, 3, 1, 0 bit 0 (freq 3072 each): 154624 -> 50*4096 + 1024 = 205824 -> 67*4096 + 0 = 274432
                                                 -> 89*4096 + 1024 = 365568 -> 119*4096 + 0 = 487424 -> 158*4096 + 2048 = 649216
    never reaches freq << 20, so no word is emitted: lane 0's state = 649216 = 0x0009E800, the other 63 lanes 0x00010000,
    and every lane's run of words has length 0: the chunk is 128 state words + 64 lengths = 192 words."""
    blob = oracle.octree_encode(np.array([[3, -4, 5]], np.int32), 32768, version=2)
    p0 = [2048] * CTX
    for k in (0, 1, 3, 6, 10, 22, 29):
        p0[k] = 1024
    p0[15] = 3072
    states = [0xE800, 0x0009] + [0x0000, 0x0001] * 63
    body = struct.pack("<I", 1) + struct.pack("<II", 4, 1) + struct.pack("<108H", *p0) + struct.pack("<I", 192)
    body += struct.pack("<128H", *states) + struct.pack("<64H", *([0] * 64))
    want = bytes([ord("O"), 2, 1, 0]) + struct.pack("<I", 1) + struct.pack("<3i", 2, -4, 4) + struct.pack("<I", len(body)) + body
    assert blob == want
    assert np.array_equal(oracle.octree_decode(blob), [[3, -4, 5]])


def test_oracle_v2_at_c3_size(oracle, wl):
    """the sweep of BASELINE.json configs[2]: the rule picks version 2, the decoded set is the input, and the blob is
    within 5 % of the serial adaptive coder's"""
    pts = wl.lidar_sweep()["points"].astype(np.int32)
    assert pts.shape[0] > oracle.OCTREE_V2_MIN_LEAVES
    blob = oracle.octree_encode(pts, 32768)
    assert blob[1] == 2
    v1 = oracle.octree_encode(pts, 32768, version=1)
    assert len(blob) <= 1.05 * len(v1), (len(blob), len(v1))
    dec = oracle.octree_decode(blob)
    assert np.array_equal(dec, oracle.octree_decode(v1))


def test_oracle_v2_rejects_corrupt_blobs(oracle):
    rng = np.random.default_rng(3)
    pts = random_cloud(rng, 3000, extent=200, lo=-100)[:, 1:]
    blob = bytearray(oracle.octree_encode(pts, 32768, version=2))
    n = struct.unpack_from("<I", blob, 4)[0]
    out = np.empty((n, 3), np.int32)

    def decode(b):
        arr = np.frombuffer(bytes(b), np.uint8)
        return oracle.lib.orc_octree_decode(arr.ctypes.data_as(C.c_void_p), C.c_int64(len(b)), out.ctypes.data_as(C.c_void_p),
                                            C.c_int64(n))
    assert decode(blob) == n
    depth = blob[2]
    head = 24 + 4 * depth + 8 + 216
    for off in (24, 24 + 4 * (depth - 1), 24 + 4 * depth, 24 + 4 * depth + 4, 24 + 4 * depth + 8, head, head + 4 + 2 * 128 + 5):
        bad = bytearray(blob)                       # level sizes, S, chunk count, p0, chunk table, a lane's word count
        bad[off] ^= 0x40
        assert decode(bad) == -1, off
    errors = 0
    for off in range(len(blob) - 900, len(blob), 11):   # renormalisation words: an error, or (a flip that leaves the child
        bad = bytearray(blob)                           # counts alone) another set of the same size — never a crash
        bad[off] ^= 0x40
        rc = decode(bad)
        assert rc in (-1, n), off
        errors += rc == -1
    assert errors > 40
    assert decode(blob[:-2]) == -1


# ------------------------------------------------------------------ GPU
def _sorted_keys(oracle, pts):
    c = np.concatenate([np.zeros((len(pts), 1), np.int32), np.asarray(pts, np.int32)], 1)
    return np.sort(oracle.morton_keys(c))


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n, _ in _clouds()])
def test_hip_v2_small_sets_forced(rt, oracle, name):
    pts = np.unique(dict(_clouds())[name].astype(np.int32), axis=0)
    keys = _sorted_keys(oracle, pts)
    kd = rt.to_device(keys.view(np.int64))
    blob = rt.octree_encode(kd, 0, version=2)
    assert blob == oracle.octree_encode(pts, 32768, version=2)
    assert np.array_equal(rt.octree_decode(blob), oracle.octree_decode(blob))
    assert rt.octree_encode(kd, 0, version=1) == oracle.octree_encode(pts, 32768, version=1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,extent,shift", [(60000, 6000, 0), (65536, 48, 0), (65537, 48, 0), (200_000, 70, 0), (150_000, 9000, 0),
                                            (70_000, 300, 9)])
def test_hip_v2_sizes(rt, oracle, n, extent, shift):
    """both forms of the level builder (one workgroup up to 65536 leaves, one wave per 64 leaves above), dense and
    scattered sets (scattered: up to ten nodes per leaf — the coder's scratch leaves the arena), stride-8 keys"""
    utils = pkg("utils")
    rng = np.random.default_rng(n)
    if extent ** 3 < 4 * n:
        g = np.stack(np.meshgrid(*[np.arange(extent)] * 3, indexing="ij"), -1).reshape(-1, 3)
        g = g[rng.permutation(len(g))[:n]] - extent // 2
    else:
        g = np.unique(rng.integers(-extent // 2, extent // 2, (2 * n, 3)), axis=0)
        g = g[rng.permutation(len(g))[:n]]
    stride = 1 << (shift // 3)
    keys = _sorted_keys(oracle, g * stride)
    kd = rt.to_device(keys.view(np.int64))
    bias = 32768 // stride
    for version in (2, 1):
        blob = rt.octree_encode(kd, shift, version=version)
        assert blob == oracle.octree_encode(g, bias, version=version), version
        dec = rt.octree_decode(blob)
        assert np.array_equal(dec * stride, oracle.keys_to_coords(keys)[:, 1:]), version
    # the operator picks the version by the leaf count, like the oracle
    blob = utils.gpcc_encode(kd, keys.view(np.int64), 0, len(keys), shift)
    assert blob[1] == (2 if n > 65536 else (3 if n >= 8192 else 1)) and blob == oracle.octree_encode(g, bias)
    assert np.array_equal(utils.gpcc_decode(blob, stride), oracle.keys_to_coords(keys)[:, 1:])


@pytest.mark.gpu
def test_hip_v2_decode_leaves_points_on_the_device(rt, oracle, wl):
    import torch
    runtime = pkg("runtime")
    pts = wl.lidar_sweep(32, 900, seed=3)["points"].astype(np.int32)
    blob = oracle.octree_encode(pts, 32768, version=2)
    buf = np.frombuffer(blob, np.uint8)
    n = C.c_int64(0)
    level_n = (C.c_int64 * 16)()
    runtime.check(rt.lib.pcc_octree_decode_dev(rt.ctx, buf.ctypes.data, len(blob), None, 0, C.byref(n), level_n), "peek")
    assert n.value == len(np.unique(pts, axis=0))
    d = torch.empty((n.value, 3), dtype=torch.int32, device="cuda")
    runtime.check(rt.lib.pcc_octree_decode_dev(rt.ctx, buf.ctypes.data, len(blob), C.c_void_p(d.data_ptr()), n.value, C.byref(n),
                                               level_n), "pcc_octree_decode_dev")
    assert np.array_equal(d.cpu().numpy(), oracle.octree_decode(blob))
    depth = blob[2]
    assert [int(v) for v in level_n][:depth] == list(struct.unpack_from("<%dI" % depth, blob, 24)) and (depth == 16 or level_n[depth] == 0)
    # version 1 through the same call
    b1 = oracle.octree_encode(pts, 32768, version=1)
    buf1 = np.frombuffer(b1, np.uint8)
    runtime.check(rt.lib.pcc_octree_decode_dev(rt.ctx, buf1.ctypes.data, len(b1), C.c_void_p(d.data_ptr()), n.value, C.byref(n),
                                               level_n), "pcc_octree_decode_dev")
    assert np.array_equal(d.cpu().numpy(), oracle.octree_decode(b1))
    # the host-only decoder refuses version 2 and says what to call
    small = np.empty((n.value, 3), np.int32)
    assert rt.lib.pcc_octree_decode(buf.ctypes.data, len(blob), small.ctypes.data, n.value, C.byref(n)) == -5
    assert b"pcc_octree_decode_ctx" in rt.lib.pcc_last_error()


@pytest.mark.gpu
def test_hip_v2_corrupt_blobs_are_errors(rt, oracle):
    """every header field and bytes of the payload flipped: an error code or (a flip the coder cannot see) a decoded
    set — never a crash, never more points than announced"""
    runtime = pkg("runtime")
    rng = np.random.default_rng(8)
    pts = random_cloud(rng, 5000, extent=300, lo=-150)[:, 1:]
    blob = oracle.octree_encode(pts, 32768, version=2)
    n = struct.unpack_from("<I", blob, 4)[0]
    out = np.empty((n, 3), np.int32)
    cnt = C.c_int64(0)

    def decode(b):
        buf = np.frombuffer(bytes(b), np.uint8)
        return rt.lib.pcc_octree_decode_ctx(rt.ctx, buf.ctypes.data, len(b), out.ctypes.data, n, C.byref(cnt))
    assert decode(blob) == 0 and np.array_equal(out, oracle.octree_decode(blob))
    depth = blob[2]
    errors = 0
    offs = list(range(0, 24 + 4 * depth + 8 + 216 + 4)) + list(range(len(blob) - 600, len(blob), 7))
    for off in offs:
        for bit in (0x01, 0x80):
            bad = bytearray(blob)
            bad[off] ^= bit
            rc = decode(bad)
            assert rc in (0, -2, -5, -6), (off, bit, rc)
            errors += rc != 0
    assert errors > len(offs)             # most flips are caught
    assert decode(blob[:-2]) == -5 and decode(blob[:40]) == -5
    assert decode(blob) == 0 and np.array_equal(out, oracle.octree_decode(blob))
