"""GPU-side interleaved rANS (container version 1; csrc/rans_gpu.hip) against its sequential restatement in the
oracle (orc_rans_interleaved_*): same bytes, same symbols, both directions, through the C-ABI.

The coding step is the reference coder's (compressai.ans, called at sender/encoder/codec_pipeline.py:305-306,426-430
and receiver/decoder/codec_parallel.py:307,398-400); what differs is the dealing of the symbols to 64 states per
wave, which only this build's decoders read (flagged container)."""
import struct

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _tables(oracle, which):
    return oracle._tables(which)


@pytest.fixture(scope="module")
def coders(oracle):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    runtime = pkg("runtime")
    out = {w: runtime.RansDev(*_tables(oracle, w)) for w in ("gaussian_conditional", "entropy_bottleneck")}
    yield out
    for c in out.values():
        c.close()


def _gaussian_case(oracle, rng, n, escapes=0.0):
    cdf, sizes, offs = _tables(oracle, "gaussian_conditional")
    idx = rng.integers(0, cdf.shape[0], n).astype(np.uint8)
    half = (sizes[idx.astype(np.int64)] - 2) // 2
    sym = np.rint(rng.normal(0, 1, n) * np.maximum(half, 1) * 0.4).astype(np.int32)
    if escapes > 0 and n:
        k = rng.random(n) < escapes
        sym[k] = rng.integers(-70000, 70000, int(k.sum()))
        sym[rng.integers(0, n, 3)] = [2 ** 31 - 1000, -2 ** 31 + 1000, 4095][:3]     # 8-nibble escapes
    return sym, idx


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 127, 4096, 32768, 32769, 65536, 100001, 262144, 262145, 300000])
def test_bytes_equal_oracle_and_round_trip(rt, oracle, coders, n):
    rng = np.random.default_rng(n)
    sym, idx = _gaussian_case(oracle, rng, n, escapes=0.01)
    gc = coders["gaussian_conditional"]
    want = oracle.rans_interleaved_encode(sym, idx, "gaussian_conditional")
    got = gc.encode(rt, rt.to_device(sym.reshape(1, -1)), rt.to_device(idx.reshape(1, -1)))[0]
    assert got == want
    magic, hn, steps, chunks = struct.unpack_from("<4sIII", got, 0)
    assert magic == b"PCI2" and hn == n
    assert steps == (320 if n > 262144 else 80 if n > 32768 else max((n + 63) // 64, 1))
    assert chunks == max(-(-n // (64 * steps)), 1)
    assert np.array_equal(oracle.rans_interleaved_decode(got, idx, n, "gaussian_conditional"), sym)
    back = gc.decode(rt, got, n, rt.to_device(idx) if n else None, 1)
    assert np.array_equal(back.cpu().numpy(), sym)


def test_streams_of_a_batch_and_channel_major_tables(rt, oracle, coders):
    """Q quality streams in one call; the factorized bottleneck's [C, N] array takes table c for channel c"""
    rng = np.random.default_rng(3)
    n = 40000
    syms, idxs = zip(*[_gaussian_case(oracle, rng, n) for _ in range(3)])
    got = coders["gaussian_conditional"].encode(rt, rt.to_device(np.stack(syms)), rt.to_device(np.stack(idxs)))
    assert got == [oracle.rans_interleaved_encode(s, i, "gaussian_conditional") for s, i in zip(syms, idxs)]
    c, nz = 32, 777
    z = rng.integers(-9, 10, (c, nz)).astype(np.int32)
    eb = coders["entropy_bottleneck"]
    zs = eb.encode(rt, rt.to_device(z.reshape(1, -1)), None, nz)[0]
    assert zs == oracle.rans_interleaved_encode(z, None, "entropy_bottleneck", idx_run=nz)
    assert np.array_equal(eb.decode(rt, zs, c * nz, None, nz).cpu().numpy().reshape(c, nz), z)
    assert np.array_equal(oracle.rans_interleaved_decode(zs, None, c * nz, "entropy_bottleneck", idx_run=nz).reshape(c, nz), z)


def test_escape_heavy_input_takes_the_large_buffer(rt, oracle, coders):
    """every symbol far outside its table: ten coding rounds per symbol, the encoder's second attempt"""
    rng = np.random.default_rng(11)
    n = 5000
    idx = rng.integers(0, 8, n).astype(np.uint8)
    sym = rng.integers(-2 ** 30, 2 ** 30, n).astype(np.int32)
    gc = coders["gaussian_conditional"]
    got = gc.encode(rt, rt.to_device(sym.reshape(1, -1)), rt.to_device(idx.reshape(1, -1)))[0]
    assert got == oracle.rans_interleaved_encode(sym, idx, "gaussian_conditional")
    assert np.array_equal(gc.decode(rt, got, n, rt.to_device(idx), 1).cpu().numpy(), sym)


def test_rate_is_close_to_the_single_stream(rt, oracle, coders):
    """the 64 states of a chunk cost 256 B (chunks of 64 x 256 symbols); on a 1M-point-frame-sized array that is a few percent"""
    rng = np.random.default_rng(5)
    n = 32 * 26386
    sym, idx = _gaussian_case(oracle, rng, n)
    inter = coders["gaussian_conditional"].encode(rt, rt.to_device(sym.reshape(1, -1)), rt.to_device(idx.reshape(1, -1)))[0]
    single = oracle.rans_encode(sym, idx.astype(np.int32), "gaussian_conditional")
    assert len(single) < len(inter) < 1.07 * len(single) + 1024


def test_malformed_streams_are_refused(rt, oracle, coders):
    runtime = pkg("runtime")
    rng = np.random.default_rng(8)
    n = 70000
    sym, idx = _gaussian_case(oracle, rng, n)
    gc = coders["gaussian_conditional"]
    good = gc.encode(rt, rt.to_device(sym.reshape(1, -1)), rt.to_device(idx.reshape(1, -1)))[0]
    d_idx = rt.to_device(idx)
    for bad in (good[:-4], good[:15], b"XXXX" + good[4:], good + b"\0\0\0\0",
                good[:8] + struct.pack("<I", 7) + good[12:],                       # steps that do not cover n
                good[:16] + struct.pack("<I", 5) + good[20:]):                     # a chunk without its states
        with pytest.raises(runtime.PccError):
            gc.decode(rt, bad, n, d_idx, 1)
    # payload corruption inside a chunk: decodes to something, or is reported — never a fault
    rng2 = np.random.default_rng(9)
    for _ in range(10):
        b = bytearray(good)
        pos = int(rng2.integers(16 + 4 * 3, len(b)))
        b[pos] ^= int(rng2.integers(1, 256))
        try:
            out = gc.decode(rt, bytes(b), n, d_idx, 1)
            assert out.shape[0] == n
        except runtime.PccError as e:
            assert e.code == -5
    assert np.array_equal(gc.decode(rt, good, n, d_idx, 1).cpu().numpy(), sym)


# ---------------------------------------------------------------------------------- whole containers, version 1
SETTINGS = [[1.0, 0.0], [0.0, 1.0], [1, 1]]


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(x["points"], y["points"]) and np.array_equal(x["colors"], y["colors"])
                                    for x, y in zip(a, b))


@pytest.fixture(scope="module")
def pipes():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    p = pkg()
    return (p.CompressionPipeline(SETTINGS, slots=1), p.CompressionPipeline(SETTINGS, slots=1, container_version=1),
            p.DecompressionPipeline(slots=1))


def test_version1_containers_equal_oracle_and_decode_like_version0(pipes, oracle, wl):
    """the flagged container: same fields, y / z strings from the GPU coder.  Bytes equal the oracle's version-1
    containers; the decoder reads both versions and reconstructs the same frames from either (same symbols)"""
    enc0, enc1, dec = pipes
    frames = [wl.sphere_shell(40, 14.7, seed=3, offset=(-60, 11, -25)), wl.room(90_000, seed=2, extent=(128, 128, 64)),
              wl.sphere_shell(24, 9.1, seed=5)]
    out0, _ = enc0.compress(wl.gop([dict(f) for f in frames]))
    out1, side1 = enc1.compress(wl.gop([dict(f) for f in frames]))
    ref1, _ = oracle.compress([dict(f) for f in frames], SETTINGS, version=1)
    for q in (1, 2, 3):
        assert out1[q] == ref1[q], f"version-1 container {q} differs from the oracle"
        assert out1[q][0] == 1 and out0[q][0] == 0                      # the flag: top byte of the first word
        assert out1[q][1:20] == out0[q][1:20]                           # frame count, q_g, q_a unchanged
        assert struct.unpack_from(">ii", out1[q], 20) == struct.unpack_from(">ii", out0[q], 20)   # N_y, N_z
        rec1, _ = dec.decompress(out1[q])
        rec0, _ = dec.decompress(out0[q])
        assert _same(rec1, rec0)
        assert _same(rec1, oracle.decompress(ref1[q]))
    assert len(side1["gop_info"]["bpp"]) == 4
    # the rate price of 64 states per chunk
    assert len(out0[3]) < len(out1[3]) < 1.12 * len(out0[3]) + 2048


def test_version1_full_size_bench_frame_equals_oracle(pipes, oracle, wl):
    enc0, enc1, dec = pipes
    frame = wl.room(1_000_000, seed=0)
    out1, _ = enc1.compress(wl.gop([dict(frame)]))
    ref1, _ = oracle.compress([dict(frame)], SETTINGS, version=1)
    assert all(out1[q] == ref1[q] for q in (1, 2, 3))
    rec1, _ = dec.decompress(out1[3])
    assert _same(rec1, oracle.decompress(ref1[3]))
    out0, _ = enc0.compress(wl.gop([dict(frame)]))
    assert _same(rec1, dec.decompress(out0[3])[0])
    assert len(out1[3]) < 1.08 * len(out0[3])


def test_version1_corrupted_containers_never_fault(pipes, wl):
    enc0, enc1, dec = pipes
    native = pkg("native")
    out1, _ = enc1.compress(wl.gop([wl.sphere_shell(32, 11.2, seed=8), wl.sphere_shell(20, 7.5, seed=4)]))
    clean = out1[3]
    n_ref = sum(f["points"].shape[0] for f in dec.decompress(clean)[0])
    rng = np.random.default_rng(77)
    outcomes = {"ok": 0, "error": 0}
    for _ in range(60):
        b = bytearray(clean)
        pos = int(rng.integers(0, len(b)))
        b[pos] ^= int(rng.integers(1, 256))
        try:
            rec, _ = dec.decompress(bytes(b))
            outcomes["ok"] += 1
        except native.PccError as e:
            assert e.code < 0
            outcomes["error"] += 1
    assert outcomes["ok"] + outcomes["error"] == 60 and outcomes["error"] > 0
    assert sum(f["points"].shape[0] for f in dec.decompress(clean)[0]) == n_ref
    # an unknown version is refused
    with pytest.raises(native.PccError):
        dec.decompress(bytes([2]) + clean[1:])


def test_version1_through_the_op_by_op_engine(pipes, wl):
    """engine="ops" (the reference's stage methods one by one in Python) writes and reads version 1 as well: the
    same bytes as the native engine, the same frames from either container"""
    p = pkg()
    enc0, enc1, dec = pipes
    enc_o = p.CompressionPipeline(SETTINGS, slots=1, engine="ops", container_version=1)
    dec_o = p.DecompressionPipeline(slots=1, engine="ops")
    frames = [wl.sphere_shell(32, 11.2, seed=8, offset=(30, -70, 5)), wl.sphere_shell(24, 9.1, seed=2)]
    out_o, _ = enc_o.compress(wl.gop([dict(f) for f in frames]))
    out_n, _ = enc1.compress(wl.gop([dict(f) for f in frames]))
    assert all(out_o[q] == out_n[q] for q in (1, 2, 3))
    assert _same(dec_o.decompress(out_o[2])[0], dec.decompress(out_n[2])[0])
    out0, _ = enc0.compress(wl.gop([dict(f) for f in frames]))
    assert _same(dec_o.decompress(out0[2])[0], dec.decompress(out_n[2])[0])
