"""Seek points of the host-coded y strings (include/pcc.h: pcc_rans_encode_seek / pcc_rans_decode_range,
pcc_codec_set_seek_points): the reference's container with a trailer behind its last frame record that lets this
library decode the y string on several host threads.  CPU part: the host coder against the oracle and against itself
(pieces from the points == the serial decode); GPU part: containers against the oracle's, the prefix property, corrupt
trailers."""
import ctypes as C
import struct

import numpy as np
import pytest

from conftest import pkg

SETTINGS = [[1.0, 0.0], [0.0, 1.0], [1, 1]]


def _stream(oracle, n, seed, spread=3.0):
    rng = np.random.default_rng(seed)
    cdf, sizes, offs = oracle._tables("gaussian_conditional")
    idx = rng.integers(0, cdf.shape[0], n).astype(np.int32)
    sym = np.rint(rng.normal(0, spread, n)).astype(np.int32)
    sym[rng.integers(0, n, max(n // 500, 1))] = rng.integers(-40000, 40000, max(n // 500, 1))     # escapes, both signs
    return sym, idx, cdf, sizes, offs


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _encode_seek(lib, sym, idx, cdf, sizes, offs, seek_index):
    si = np.asarray(seek_index, dtype=np.int64)
    st, wd = np.zeros(max(len(si), 1), np.uint64), np.zeros(max(len(si), 1), np.int64)
    out = np.empty(8 * len(sym) + 64, np.uint8)
    ln = C.c_int64(0)
    rc = lib.pcc_rans_encode_seek(_p(sym), _p(idx), len(sym), _p(cdf), cdf.shape[1], _p(sizes), _p(offs), cdf.shape[0], _p(out),
                                  out.shape[0], C.byref(ln), _p(si) if len(si) else None, len(si), _p(st), _p(wd))
    assert rc == 0, lib.pcc_last_error()
    return out[:ln.value].tobytes(), st[:len(si)], wd[:len(si)]


@pytest.mark.parametrize("n,points", [(5000, [64, 1000, 1001, 4999]), (200_000, [12480, 100_032, 150_016]), (300, [0, 150, 300, 400])])
def test_seek_points_of_the_host_coder(oracle, n, points):
    """the product's host coder and the oracle's record the same state / position at every seek point, the stream is the
    plain encoder's, and decoding the pieces from the points (any order) gives the serial decode and ends every piece
    exactly on the next point"""
    lib = pkg("_abi").lib()
    sym, idx, cdf, sizes, offs = _stream(oracle, n, n)
    stream, st, wd = _encode_seek(lib, sym, idx, cdf, sizes, offs, points)
    assert stream == oracle.rans_encode(sym, idx, "gaussian_conditional")
    o_st, o_wd = np.zeros(len(points), np.uint64), np.zeros(len(points), np.int64)
    out = np.empty(8 * n + 64, np.uint8)
    si = np.asarray(points, np.int64)
    oracle.lib.orc_rans_encode_seek.restype = C.c_int64
    got = oracle.lib.orc_rans_encode_seek(_p(sym), _p(idx), C.c_int64(n), _p(cdf), C.c_int(cdf.shape[1]), _p(sizes), _p(offs), _p(out),
                                          C.c_int64(out.shape[0]), _p(si), C.c_int(len(points)), _p(o_st), _p(o_wd))
    assert out[:got].tobytes() == stream
    assert np.array_equal(st, o_st) and np.array_equal(wd, o_wd)
    inside = [k for k, i in enumerate(points) if 0 < i < n]
    assert all(st[k] >= (1 << 31) and 2 <= wd[k] <= len(stream) // 4 for k in inside)
    assert all(st[k] == 0 and wd[k] == 0 for k in range(len(points)) if k not in inside)
    # pieces, last first
    buf = np.frombuffer(stream, np.uint8)
    dec = np.full(n, -(1 << 30), np.int32)
    cuts = [0] + [points[k] for k in inside] + [n]
    pos = [(0, 0)] + [(int(st[k]), int(wd[k])) for k in inside]
    ends = {}
    for j in reversed(range(len(cuts) - 1)):
        xo, wo = C.c_uint64(0), C.c_int64(0)
        rc = lib.pcc_rans_decode_range(_p(buf), len(stream), _p(idx), n, _p(cdf), cdf.shape[1], _p(sizes), _p(offs), cdf.shape[0],
                                       _p(dec), cuts[j], cuts[j + 1], C.c_uint64(pos[j][0]), pos[j][1], C.byref(xo), C.byref(wo))
        assert rc == 0, lib.pcc_last_error()
        ends[j] = (xo.value, wo.value)
    assert np.array_equal(dec, sym)
    for j in range(len(cuts) - 2):
        assert ends[j] == pos[j + 1], j
    assert ends[len(cuts) - 2][1] == len(stream) // 4            # the last piece ends with the stream
    # a piece started from a wrong state decodes something else and does not land on the next point
    if len(cuts) > 2:
        xo, wo = C.c_uint64(0), C.c_int64(0)
        lib.pcc_rans_decode_range(_p(buf), len(stream), _p(idx), n, _p(cdf), cdf.shape[1], _p(sizes), _p(offs), cdf.shape[0],
                                  _p(dec), cuts[1], cuts[2], C.c_uint64(pos[1][0] ^ 0x10), pos[1][1], C.byref(xo), C.byref(wo))
        assert (xo.value, wo.value) != (pos[2] if len(pos) > 2 else (0, len(stream) // 4))


def test_seek_point_known_answer(oracle):
    """Worked by hand.  One table, cdf = [0, 16384, 32768, 65536]: symbols 0 and 1 with frequency 2^14 each, the last
    bin (2^15) the escape bin.  Symbols [0, 1, 0, 1], one seek point at index 2.  rANS64 (L = 2^31, 16-bit precision),
    coded last symbol first; a step is x -> ((x / f) << 16) + x % f + start, and no state here reaches the
    renormalisation threshold ((L >> 16) << 32) f = 2^61:
      i = 3, s = 1 (start 2^14): x = (2^31 >> 14 << 16) + 0 + 2^14            = 2^33 + 2^14
      i = 2, s = 0 (start 0)   : x = ((2^33 + 2^14) >> 14 << 16) + 0          = 2^35 + 2^16     <- symbols >= 2 are coded
      i = 1, s = 1             : x = ((2^35 + 2^16) >> 14 << 16) + 0 + 2^14   = 2^37 + 2^18 + 2^14
      i = 0, s = 0             : x = ((2^37 + 2^18 + 2^14) >> 14 << 16) + 0   = 2^39 + 2^20 + 2^16
    stream = the final state, low word first (8 bytes, no renormalisation word).  A decoder has consumed those 2 words
    when symbol 2 is next, and its state there is 2^35 + 2^16; decoding [2, 4) from the point gives [0, 1] and ends on
    the coder's initial state 2^31 with still 2 words consumed."""
    lib = pkg("_abi").lib()
    cdf = np.array([[0, 16384, 32768, 65536]], np.int32)
    sizes, offs = np.array([4], np.int32), np.array([0], np.int32)
    sym, idx = np.array([0, 1, 0, 1], np.int32), np.zeros(4, np.int32)
    stream, st, wd = _encode_seek(lib, sym, idx, cdf, sizes, offs, [2])
    x_final = (1 << 39) + (1 << 20) + (1 << 16)
    assert stream == struct.pack("<II", x_final & 0xFFFFFFFF, x_final >> 32)
    assert int(st[0]) == (1 << 35) + (1 << 16) and int(wd[0]) == 2
    o_st, o_wd = np.zeros(1, np.uint64), np.zeros(1, np.int64)
    out = np.empty(64, np.uint8)
    si = np.array([2], np.int64)
    oracle.lib.orc_rans_encode_seek.restype = C.c_int64
    got = oracle.lib.orc_rans_encode_seek(_p(sym), _p(idx), C.c_int64(4), _p(cdf), C.c_int(4), _p(sizes), _p(offs), _p(out), C.c_int64(64),
                                          _p(si), C.c_int(1), _p(o_st), _p(o_wd))
    assert out[:got].tobytes() == stream and int(o_st[0]) == int(st[0]) and int(o_wd[0]) == 2
    dec = np.full(4, -9, np.int32)
    xo, wo = C.c_uint64(0), C.c_int64(0)
    buf = np.frombuffer(stream, np.uint8)
    rc = lib.pcc_rans_decode_range(_p(buf), len(stream), _p(idx), 4, _p(cdf), 4, _p(sizes), _p(offs), 1, _p(dec), 2, 4,
                                   C.c_uint64(int(st[0])), 2, C.byref(xo), C.byref(wo))
    assert rc == 0 and dec.tolist() == [-9, -9, 0, 1] and xo.value == 1 << 31 and wo.value == 2
    rc = lib.pcc_rans_decode_range(_p(buf), len(stream), _p(idx), 4, _p(cdf), 4, _p(sizes), _p(offs), 1, _p(dec), 0, 2,
                                   C.c_uint64(0), 0, C.byref(xo), C.byref(wo))
    assert rc == 0 and dec.tolist() == [0, 1, 0, 1] and xo.value == int(st[0]) and wo.value == 2


def test_trailer_is_behind_the_reference_container(oracle, wl):
    """the oracle: with seek points the container is the plain container + trailer, and its reader — the reference's
    reader, codec_parallel.py:173-216 — stops in front of the trailer: same reconstruction"""
    frames = [wl.room(150_000, seed=5, extent=(256, 256, 128))]
    plain, _ = oracle.compress([dict(f) for f in frames], SETTINGS)
    seek, _ = oracle.compress([dict(f) for f in frames], SETTINGS, seek_points=8)
    for q in (1, 2, 3):
        assert seek[q][:len(plain[q])] == plain[q]
        tr = seek[q][len(plain[q]):]
        assert tr[:4] == b"PCSK" and len(tr) == 8 + 16 * struct.unpack(">i", tr[4:8])[0]
    a, b = oracle.decompress(plain[3]), oracle.decompress(seek[3])
    assert all(np.array_equal(x["points"], y["points"]) and np.array_equal(x["colors"], y["colors"]) for x, y in zip(a, b))
    small, _ = oracle.compress([wl.sphere_shell(24, 9.1, seed=2)], SETTINGS, seek_points=8)
    assert b"PCSK" not in small[1][-16:]                         # short strings carry no points


@pytest.mark.gpu
def test_hip_containers_with_seek_points(oracle, wl):
    """CompressionPipeline(seek_points=8): containers equal the oracle's byte for byte; decoding them (threads), the plain
    containers (serial) and containers whose trailer is damaged (fallback to serial) all give the oracle's frames"""
    frames = [wl.room(150_000, seed=5, extent=(256, 256, 128)), wl.room(90_000, seed=6, extent=(256, 256, 128))]
    ref, _ = oracle.compress([dict(f) for f in frames], SETTINGS, seek_points=8)
    plain, _ = oracle.compress([dict(f) for f in frames], SETTINGS)
    enc = pkg("codec_pipeline").CompressionPipeline(SETTINGS, slots=1, seek_points=8)
    enc0 = pkg("codec_pipeline").CompressionPipeline(SETTINGS, slots=1)
    dec = pkg("codec_parallel").DecompressionPipeline(slots=1)
    out, _ = enc.compress(wl.gop([dict(f) for f in frames]))
    out0, _ = enc0.compress(wl.gop([dict(f) for f in frames]))
    want = oracle.decompress(ref[3])

    def same(rec):
        return len(rec) == len(want) and all(np.array_equal(a["points"], b["points"]) and np.array_equal(a["colors"], b["colors"])
                                             for a, b in zip(rec, want))
    for q in (1, 2, 3):
        assert out[q] == ref[q] and out0[q] == plain[q] and out[q][:len(out0[q])] == out0[q]
    assert same(dec.decompress(out[3])[0]) and same(dec.decompress(out0[3])[0])
    n_tr = len(out[3]) - len(out0[3])
    assert n_tr == 8 + 16 * 7
    for off in range(len(out0[3]), len(out[3]), 3):               # every third byte of the trailer damaged
        bad = bytearray(out[3])
        bad[off] ^= 0x21
        assert same(dec.decompress(bytes(bad))[0]), off
    # the op-by-op engine reads the container field by field like the reference's read_bitstream_batched and never looks
    # behind the last frame record
    dec_ops = pkg("codec_parallel").DecompressionPipeline(slots=1, engine="ops")
    assert same(dec_ops.decompress(out[3])[0])
    assert same(dec.decompress(out[3][:-5])[0])                    # a cut trailer
    assert same(dec.decompress(out[3] + b"tail")[0])               # bytes behind it
    with pytest.raises(ValueError):
        pkg("codec_pipeline").CompressionPipeline(SETTINGS, slots=1, seek_points=8, container_version=1)
