"""Whole-GOP C entry points (pcc_encode_gop / pcc_decode_gop) against the op-by-op Python
mirror of the reference's stage methods and against the CPU oracle: containers byte for byte,
reconstructions bit for bit."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SETTINGS = [[1.0, 0.0], [0.0, 1.0], [1, 1]]


def _stack(frames):
    pts = np.concatenate([np.concatenate([np.full((f["points"].shape[0], 1), i), f["points"].astype(np.int64)], 1)
                          for i, f in enumerate(frames)], 0).astype(np.int32)
    col = np.concatenate([f["colors"] for f in frames], 0).astype(np.float32)
    feats = np.concatenate([np.ones((col.shape[0], 1), np.float32), col], 1)
    return torch.from_numpy(pts).cuda(), torch.from_numpy(np.ascontiguousarray(feats)).cuda()


@pytest.fixture(scope="module")
def codec():
    native = pkg("native")
    model = pkg("model")
    c = native.NativeCodec(model.load_checkpoint("demo_small"), 0)
    yield c
    c.close()


def _gops(wl):
    rng = np.random.default_rng(3)
    g = {
        "sphere2": [wl.sphere_shell(32, 11.2, seed=1, offset=(-40, 8, -90)), wl.sphere_shell(24, 9.1, seed=2)],
        "sphere1": [wl.sphere_shell(40, 15.0, seed=5)],
        "body3": [wl.body(20000, seed=s) for s in (1, 2, 3)],
        "lidar": [wl.lidar_sweep(16, 600, seed=4)],
    }
    return g


@pytest.mark.parametrize("name", ["sphere2", "sphere1", "body3", "lidar"])
def test_native_gop_equals_python_ops_and_oracle(wl, oracle, codec, name):
    frames = _gops(wl)[name]
    enc = pkg("codec_pipeline").CompressionPipeline(SETTINGS, slots=1, engine="ops")
    dec = pkg("codec_parallel").DecompressionPipeline(slots=1, engine="ops")
    out, side = enc.compress(wl.gop([dict(f) for f in frames]))
    coords, feats = _stack(frames)
    cont, ks, times = codec.encode(coords, feats, len(frames), SETTINGS)
    for q in range(len(SETTINGS)):
        assert cont[q] == out[q + 1], f"container {q + 1} differs from the op-by-op path"
    assert set(times) == set(side["enc_time_measurements"])
    # the per-frame entry point (frames as the capturer leaves them: int16 points, float64 colours)
    cont_f, ks_f, _ = codec.encode_frames([torch.from_numpy(np.ascontiguousarray(f["points"])).cuda() for f in frames],
                                          [torch.from_numpy(np.ascontiguousarray(f["colors"])).cuda() for f in frames],
                                          SETTINGS)
    assert cont_f == cont and ks_f == ks
    ref, _ = oracle.compress(frames, SETTINGS)
    for q in range(len(SETTINGS)):
        assert cont[q] == ref[q + 1], f"container {q + 1} differs from the oracle"
    # decode: native vs op-by-op path vs oracle
    for q in (1, 3):
        rec, _ = dec.decompress(out[q])
        c, col, offs, qq, dts = codec.decode(cont[q - 1])
        assert qq == [float(v) for v in SETTINGS[q - 1]]
        assert len(offs) - 1 == len(rec)
        ch, colh = c.cpu().numpy(), col.cpu().numpy()
        oref = oracle.decompress(ref[q])
        for i, fr in enumerate(rec):
            assert np.array_equal(ch[offs[i]:offs[i + 1], 1:], fr["points"])
            assert np.all(ch[offs[i]:offs[i + 1], 0] == i)
            item = np.clip(np.nan_to_num(colh[offs[i]:offs[i + 1]], nan=0.0) * 255.0, 0, 255) / 255
            assert np.array_equal(item, fr["colors"])
            assert np.array_equal(fr["points"], oref[i]["points"]) and np.array_equal(fr["colors"], oref[i]["colors"])


def test_native_frame_tables(wl, codec):
    """pcc_encode_gop_frames: every dtype pair it reads directly, an empty frame in the middle, 32 frames (its
    limit) and the errors; the pipeline class falls back to its own stacking for mixed dtypes and longer GOPs"""
    frames = [wl.sphere_shell(24, 9.1, seed=s, offset=(3 * s, -2 * s, s)) for s in range(1, 4)]
    coords, feats = _stack(frames)
    ref, ks, _ = codec.encode(coords, feats, len(frames), SETTINGS)
    for pdt in (torch.int16, torch.int32):
        for cdt in (torch.float64, torch.float32):
            # float32 colours: the reference casts float64 -> float32 first, so feed the already-cast values
            pts = [torch.from_numpy(f["points"].astype(np.int64)).to(pdt).cuda() for f in frames]
            cols = [torch.from_numpy(f["colors"].astype(np.float32 if cdt == torch.float32 else np.float64)).cuda()
                    for f in frames]
            got, ks2, _ = codec.encode_frames(pts, cols, SETTINGS)
            assert got == ref and ks2 == ks
    native = pkg("native")
    rtm = pkg("runtime")
    # out-of-range int32 coordinate -> PCC_E_RANGE, duplicates -> PCC_E_DUP
    bad = [torch.tensor([[0, 0, 0], [40000, 0, 0]], dtype=torch.int32).cuda()]
    with pytest.raises(rtm.PccError) as e:
        codec.encode_frames(bad, [torch.zeros((2, 3)).cuda()], SETTINGS)
    assert e.value.code == -3
    dup = [torch.tensor([[1, 2, 3], [1, 2, 3]], dtype=torch.int16).cuda()]
    with pytest.raises(rtm.PccError) as e:
        codec.encode_frames(dup, [torch.zeros((2, 3)).cuda()], SETTINGS)
    assert e.value.code == -4
    # pipeline class: an empty frame inside the GOP, 32 and 33 frames, mixed dtypes — all equal to the op-by-op path
    cp = pkg("codec_pipeline")
    enc_n = cp.CompressionPipeline(SETTINGS, slots=1)
    enc_o = cp.CompressionPipeline(SETTINGS, slots=1, engine="ops")
    small = [wl.sphere_shell(12, 4.0, seed=s, offset=(s, s, -s)) for s in range(33)]
    empty = {"points": np.zeros((0, 3), np.int16), "colors": np.zeros((0, 3), np.float64)}
    mixed = [dict(frames[0]), {"points": frames[1]["points"].astype(np.int32),
                               "colors": frames[1]["colors"].astype(np.float32)}]
    for gop in ([frames[0], empty, frames[1]], small[:32], small, mixed):
        a, _ = enc_n.compress(wl.gop([dict(f) for f in gop]))
        b, _ = enc_o.compress(wl.gop([dict(f) for f in gop]))
        assert [a[q] for q in (1, 2, 3)] == [b[q] for q in (1, 2, 3)]


def test_native_symbols_beyond_int16(wl):
    """colours far outside [0,1] drive latent symbols past +-32767: the compact int16 symbol transfer reports the
    overflow and the encoder codes the GOP through the generic int32 path (escape-coded symbols) — same containers
    as the op-by-op engine, and they decode"""
    cp, dp = pkg("codec_pipeline"), pkg("codec_parallel")
    frames = [wl.sphere_shell(24, 9.1, seed=2), wl.sphere_shell(20, 7.7, seed=3, offset=(9, -4, 2))]
    for f in frames:
        f["colors"] = f["colors"] * 3.0e6 - 1.0e6
    a, _ = cp.CompressionPipeline(SETTINGS, slots=1).compress(wl.gop([dict(f) for f in frames]))
    b, _ = cp.CompressionPipeline(SETTINGS, slots=1, engine="ops").compress(wl.gop([dict(f) for f in frames]))
    assert [a[q] for q in (1, 2, 3)] == [b[q] for q in (1, 2, 3)]
    ra, _ = dp.DecompressionPipeline(slots=1).decompress(a[3])
    rb, _ = dp.DecompressionPipeline(slots=1, engine="ops").decompress(b[3])
    for x, y in zip(ra, rb):
        assert np.array_equal(x["points"], y["points"]) and np.array_equal(x["colors"], y["colors"])


def test_native_errors(codec):
    native = pkg("native")
    pts = torch.tensor([[0, 1, 2, 3], [0, 1, 2, 3]], dtype=torch.int32).cuda()
    feats = torch.ones((2, 4), dtype=torch.float32).cuda()
    with pytest.raises(native.PccError) as e:
        codec.encode(pts, feats, 1, SETTINGS)
    assert e.value.code == -4
    far = torch.tensor([[0, 40000, 0, 0]], dtype=torch.int32).cuda()
    with pytest.raises(native.PccError) as e:
        codec.encode(far, torch.ones((1, 4)).cuda(), 1, SETTINGS)
    assert e.value.code == -3
    with pytest.raises(native.PccError) as e:
        codec.decode(b"\x00" * 20)
    assert e.value.code == -5


def test_native_truncated_container(wl, codec):
    frames = [wl.sphere_shell(24, 9.1, seed=2)]
    coords, feats = _stack(frames)
    cont, _, _ = codec.encode(coords, feats, 1, [[1, 1]])
    native = pkg("native")
    for cut in (37, len(cont[0]) // 2, len(cont[0]) - 3):
        with pytest.raises(native.PccError):
            codec.decode(cont[0][:cut])
    c, col, offs, _, _ = codec.decode(cont[0])       # the codec is still usable afterwards
    assert c.shape[0] == frames[0]["points"].shape[0]


def test_native_decode_into_caller_arrays(wl, codec):
    """pcc_container_points + pcc_decode_gop_packed: the cloud lands in the caller's host arrays in one call, equal to
    pcc_decode_gop + pcc_decode_fetch_packed; a destination that is too small is refused and nothing is written;
    corrupted containers (the announced count included) give a cloud or a PccError, never a write past the arrays"""
    import ctypes as C
    native, abi = pkg("native"), pkg("_abi")
    frames = [wl.sphere_shell(24, 9.1, seed=2), wl.sphere_shell(20, 7.5, seed=4, offset=(30, -9, 4))]
    coords, feats = _stack(frames)
    cont, _, _ = codec.encode(coords, feats, 2, [[1, 1]])
    data = cont[0]
    n_ref = sum(f["points"].shape[0] for f in frames)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    cap, nf = C.c_int64(0), C.c_int32(0)
    abi.check(codec.lib.pcc_container_points(buf, len(data), C.byref(cap), C.byref(nf)), "pcc_container_points")
    assert cap.value == n_ref and nf.value == 2
    pts, cols, offs, _, _ = codec.decode(data, packed_host=True)          # the one-call form
    c4, col, offs2, _, _ = codec.decode(data)                               # the two-call form, device tensors
    assert offs == offs2 and np.array_equal(pts, c4.cpu().numpy()[:, 1:])
    assert np.array_equal(cols, np.clip(np.nan_to_num(col.cpu().numpy(), nan=0.0) * np.float32(255), 0, 255) / np.float32(255))
    small_p = np.full((n_ref - 1, 3), -7, np.int32)
    small_c = np.full((n_ref - 1, 3), -7, np.float32)
    info, ts = abi.PccCloudInfo(), (C.c_double * 6)()
    rc = codec.lib.pcc_decode_gop_packed(codec.handle, buf, len(data), C.c_void_p(small_p.ctypes.data),
                                         C.c_void_p(small_c.ctypes.data), n_ref - 1, C.byref(info), ts)
    assert rc == -1 and (small_p == -7).all() and (small_c == -7).all()
    rng = np.random.default_rng(99)
    for _ in range(30):
        b = bytearray(data)
        pos = int(rng.integers(0, len(b)))
        b[pos] ^= int(rng.integers(1, 256))
        try:
            p2, c2, o2, _, _ = codec.decode(bytes(b), packed_host=True)
            assert p2.shape == c2.shape and o2[-1] == p2.shape[0]
        except native.PccError as e:
            assert e.code < 0
    p3, _, _, _, _ = codec.decode(data, packed_host=True)                   # still usable
    assert np.array_equal(p3, pts)


def test_native_edge_gops(wl, oracle, codec):
    """ragged GOP through pcc_encode_gop / pcc_decode_gop: a one-voxel frame, frames at the int16 corners of the
    coordinate range, ten frames, and a middle frame that prunes to nothing at the coarse levels"""
    rng = np.random.default_rng(9)

    def cloud(n, lo, hi):
        p = np.unique(rng.integers(lo, hi, (n, 3)), axis=0).astype(np.int16)
        return {"points": p, "colors": rng.random((p.shape[0], 3))}

    gops = {
        "one_voxel_and_sphere": [{"points": np.array([[5, -3, 7]], np.int16), "colors": np.array([[0.2, 0.4, 0.6]])},
                                 wl.sphere_shell(24, 9.1, seed=2)],
        "corners": [cloud(3000, -32768, -32700), cloud(3000, 32700, 32767)],
        "ten_frames": [wl.sphere_shell(16 + f, 5.0 + f, seed=f, offset=(7 * f, -3 * f, f)) for f in range(10)],
        "dust": [cloud(400, -3000, 3000), wl.sphere_shell(20, 7.7, seed=3), cloud(2, -10, 10)],
    }
    settings = [[1, 1], [0.25, 0.75]]
    for name, frames in gops.items():
        coords, feats = _stack(frames)
        cont, ks, _ = codec.encode(coords, feats, len(frames), settings)
        ref, _ = oracle.compress(frames, settings)
        assert cont[0] == ref[1] and cont[1] == ref[2], name
        assert ks[2] == [f["points"].shape[0] for f in frames], name
        c, col, offs, _, _ = codec.decode(cont[1])
        oref = oracle.decompress(ref[2])
        ch, colh = c.cpu().numpy(), col.cpu().numpy()
        assert len(offs) - 1 == len(oref), name
        for i, fr in enumerate(oref):
            assert np.array_equal(ch[offs[i]:offs[i + 1], 1:], fr["points"]), name
            item = np.clip(np.nan_to_num(colh[offs[i]:offs[i + 1]], nan=0.0) * 255.0, 0, 255) / 255
            assert np.array_equal(item, fr["colors"]), name


def test_native_codec_reuse_across_sizes(wl, codec):
    """the per-codec device pool and pinned buffers are reused: big GOP, small GOP, big GOP again, same bytes"""
    big = [wl.body(120000, seed=1)]
    small = [wl.sphere_shell(16, 5.5, seed=1)]
    cb, fb = _stack(big)
    cs_, fs = _stack(small)
    a1, _, _ = codec.encode(cb, fb, 1, SETTINGS)
    s1, _, _ = codec.encode(cs_, fs, 1, SETTINGS)
    a2, _, _ = codec.encode(cb, fb, 1, SETTINGS)
    s2, _, _ = codec.encode(cs_, fs, 1, SETTINGS)
    assert a1 == a2 and s1 == s2
    n1 = codec.decode(a1[2])[0].shape[0]
    n2 = codec.decode(s1[2])[0].shape[0]
    assert n1 == big[0]["points"].shape[0] and n2 == small[0]["points"].shape[0]


def test_octree_one_call_entry_points(rt, oracle):
    """pcc_octree_encode / pcc_octree_decode (one call each) == the levels + pack / peek + unpack sequence that
    utils.gpcc_encode / gpcc_decode drive, and == the oracle's blob"""
    import ctypes as C
    from conftest import surface_cloud
    utils, rtm = pkg("utils"), pkg("runtime")
    rng = np.random.default_rng(21)
    coords = surface_cloud(rng, 5000, batches=1, stride=8)
    keys_h = np.sort(oracle.morton_keys(coords))
    keys = rt.to_device(keys_h.view(np.int64))
    blob_ref = utils.gpcc_encode(keys, keys_h, 0, len(keys_h), 9)
    cap = 64 + 16 * len(keys_h)
    out = np.empty(cap, np.uint8)
    n_out = C.c_int64(0)
    rtm.check(rt.lib.pcc_octree_encode(rt.ctx, rtm._ptr(keys), len(keys_h), 9, out.ctypes.data, cap, C.byref(n_out)),
              "pcc_octree_encode")
    blob = out[:n_out.value].tobytes()
    assert blob == blob_ref
    n_pts = C.c_int64(0)
    rtm.check(rt.lib.pcc_octree_decode(blob, len(blob), None, 0, C.byref(n_pts)), "pcc_octree_decode")
    assert n_pts.value == len(keys_h)
    pts = np.empty((n_pts.value, 3), np.int32)
    rtm.check(rt.lib.pcc_octree_decode(blob, len(blob), pts.ctypes.data, n_pts.value, C.byref(n_pts)), "pcc_octree_decode")
    assert np.array_equal(pts * 8, utils.gpcc_decode(blob, 8))
    small = np.empty((1, 3), np.int32)
    assert rt.lib.pcc_octree_decode(blob, len(blob), small.ctypes.data, 1, C.byref(n_pts)) == -6
    # empty frame
    rtm.check(rt.lib.pcc_octree_encode(rt.ctx, None, 0, 9, out.ctypes.data, cap, C.byref(n_out)), "pcc_octree_encode")
    assert out[:n_out.value].tobytes() == utils.gpcc_encode(keys[:0], keys_h[:0], 0, 0, 9)


def test_native_pool_survives_many_sizes(wl, codec):
    """GOP sizes that keep changing make the device pool add blocks; it is re-made as one block when the chain
    gets long, and results do not depend on the pool's history"""
    base = [wl.sphere_shell(20, 7.5, seed=4)]
    cb, fb = _stack(base)
    ref, _, _ = codec.encode(cb, fb, 1, [[1, 1]])
    for s in range(40):
        fr = [wl.sphere_shell(14 + (s * 7) % 40, 4.0 + (s % 9), seed=s)]
        c, f = _stack(fr)
        out, _, _ = codec.encode(c, f, 1, [[1, 1]])
        n = codec.decode(out[0])[0].shape[0]
        assert n == fr[0]["points"].shape[0]
    again, _, _ = codec.encode(cb, fb, 1, [[1, 1]])
    assert again == ref


def test_native_decode_of_an_empty_gop(rt, codec):
    """a container that announces frames without any latent point decodes to an empty cloud (or is refused with
    a stream error) — never a crash"""
    import struct
    utils, native = pkg("utils"), pkg("native")
    keys = rt.to_device(np.zeros(0, np.int64))
    blob = utils.gpcc_encode(keys, np.zeros(0, np.uint64), 0, 0, 9)
    for n_frames in (1, 3):
        body = struct.pack(">idd", n_frames, 1.0, 1.0) + struct.pack(">iiii", 0, 0, 0, 0)
        for _ in range(n_frames):
            body += struct.pack(">iiii", len(blob), 0, 0, 0) + blob
        try:
            c, col, offs, q, _ = codec.decode(body)
            assert c.shape[0] == 0 and col.shape[0] == 0
        except native.PccError as e:
            assert e.code in (-5, -1)
    # and the codec still works afterwards
    out, _, _ = codec.encode(*_stack([{"points": np.array([[1, 2, 3]], np.int16), "colors": np.array([[.1, .2, .3]])}]),
                             1, [[1, 1]])
    assert codec.decode(out[0])[0].shape[0] == 1


def test_native_decode_survives_corrupted_containers(wl, codec):
    """seeded single-byte corruptions anywhere in a container (header, y / z streams, geometry blobs, k fields):
    pcc_decode_gop either returns a cloud or a PccError, and the codec decodes the clean container afterwards"""
    native = pkg("native")
    frames = [wl.sphere_shell(20, 7.5, seed=4), wl.sphere_shell(16, 5.5, seed=5, offset=(40, 0, -8))]
    coords, feats = _stack(frames)
    cont, _, _ = codec.encode(coords, feats, 2, [[1, 1]])
    clean = cont[0]
    n_ref = codec.decode(clean)[0].shape[0]
    rng = np.random.default_rng(1234)
    outcomes = {"ok": 0, "error": 0}
    for _ in range(40):
        b = bytearray(clean)
        pos = int(rng.integers(0, len(b)))
        b[pos] ^= int(rng.integers(1, 256))
        try:
            c, col, offs, _, _ = codec.decode(bytes(b))
            assert c.shape[0] == col.shape[0] and offs[-1] == c.shape[0]
            outcomes["ok"] += 1
        except native.PccError as e:
            assert e.code < 0
            outcomes["error"] += 1
    assert outcomes["ok"] + outcomes["error"] == 40
    assert codec.decode(clean)[0].shape[0] == n_ref


def _blob_offsets(container):
    """byte offset of every frame's geometry blob inside a batched container (codec_parallel.py:173-216)"""
    import struct
    nf = struct.unpack_from(">i", container, 0)[0]
    ly, lz = struct.unpack_from(">ii", container, 28)
    pos, out = 36 + ly + lz, []
    for _ in range(nf):
        lp = struct.unpack_from(">i", container, pos)[0]
        out.append(pos + 16)
        pos += 16 + lp
    return out


def test_native_decode_survives_a_damaged_version_3_slot(wl, codec):
    """a frame whose latent takes geometry blob version 3 (parts decoded on the codec's threads): seeded corruptions inside
    the slot — envelope, length table, part headers, part payloads — give a cloud or a PccError, never a fault or a
    hang, and the codec decodes the clean container afterwards.  (The device path sizes its stride-16 / stride-32 sets
    from the parts' level counts: what the parts decoded and the order check must keep those exact.)"""
    import struct
    native = pkg("native")
    coords, feats = _stack([wl.room(400_000, seed=3)])
    cont, _, _ = codec.encode(coords, feats, 1, [[1, 1]])
    clean = cont[0]
    off = _blob_offsets(clean)[0]
    assert clean[off:off + 2] == b"O\x03"
    k = clean[off + 3]
    blob_len = 24 + struct.unpack_from("<I", clean, off + 20)[0]
    ref = codec.decode(clean)
    n_ref = ref[0].shape[0]
    rng = np.random.default_rng(77)
    outcomes = {"ok": 0, "error": 0}
    spots = [int(v) for v in rng.integers(0, 24 + 4 * k + 24, 25)] + [int(v) for v in rng.integers(0, blob_len, 45)]
    for i, rel in enumerate(spots):
        b = bytearray(clean)
        b[off + rel] ^= int(rng.integers(1, 256))
        if i % 9 == 8:      # and a swap of two length entries now and then
            a0, a1 = off + 24, off + 28
            b[a0:a0 + 4], b[a1:a1 + 4] = b[a1:a1 + 4], b[a0:a0 + 4]
        try:
            c, col, offs, _, _ = codec.decode(bytes(b))
            assert c.shape[0] == col.shape[0] and offs[-1] == c.shape[0]
            outcomes["ok"] += 1
        except native.PccError as e:
            assert e.code < 0
            outcomes["error"] += 1
    assert outcomes["ok"] + outcomes["error"] == len(spots) and outcomes["error"] > 0
    again = codec.decode(clean)
    assert again[0].shape[0] == n_ref and bool((again[0] == ref[0]).all())


def test_native_decode_refuses_a_moved_octree_origin(wl, codec):
    """every bit of the 12 origin bytes of a geometry blob (octree_host.cpp: bytes 8..19, little-endian int32 x 3):
    a flipped bit either leaves the root cube on its 2^depth grid (the cloud is translated; the decode succeeds with
    the same point count) or takes it off the grid / out of range, and then the decoder must refuse the container
    BEFORE the device path sizes its stride-16 / stride-32 sets from the octree's level counts"""
    native = pkg("native")
    frames = [wl.sphere_shell(20, 7.5, seed=4, offset=(-37, 5, 19)), wl.sphere_shell(16, 5.5, seed=5, offset=(40, 0, -8))]
    coords, feats = _stack(frames)
    cont, _, _ = codec.encode(coords, feats, 2, [[1, 1]])
    clean = cont[0]
    n_ref = codec.decode(clean)[0].shape[0]
    refused = moved = 0
    for off in _blob_offsets(clean):
        depth = clean[off + 2]
        for byte in range(8, 20):
            for bit in range(8):
                b = bytearray(clean)
                b[off + byte] ^= 1 << bit
                try:
                    c, col, offs, _, _ = codec.decode(bytes(b))
                    # only a translation by a multiple of the root cube's side can pass
                    assert 8 * (byte % 4) + bit >= depth and c.shape[0] == n_ref
                    moved += 1
                except native.PccError as e:
                    assert e.code in (-5, -3), e
                    refused += 1
    assert refused >= 2 * 3 * 3 and refused + moved == 2 * 96     # at least the low `depth` bits of every axis
    assert codec.decode(clean)[0].shape[0] == n_ref


def test_native_decode_refuses_announced_sizes_it_cannot_verify(wl, codec):
    """a header that announces far more latent rows than the streams hold is refused without the decoder sizing
    host or pinned buffers from the announcement"""
    import struct
    native = pkg("native")
    coords, feats = _stack([wl.sphere_shell(20, 7.5, seed=4)])
    cont, _, _ = codec.encode(coords, feats, 1, [[1, 1]])
    clean = bytearray(cont[0])
    off = _blob_offsets(bytes(clean))[0]
    for n_y in (1 << 30, (1 << 27) + 1, 5_000_000):
        b = bytearray(clean)
        struct.pack_into(">i", b, 20, n_y)          # N_y of the container header
        struct.pack_into(">i", b, 24, n_y)          # N_z <= N_y
        struct.pack_into("<I", b, off + 4, n_y)     # and the blob's own point count
        with pytest.raises(native.PccError):
            codec.decode(bytes(b))
    assert codec.decode(bytes(clean))[0].shape[0] == coords.shape[0]
