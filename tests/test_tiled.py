"""BASELINE.json config 5 — a 4M-point scan cut into 8 octree blocks, one GOP per rank, all-gather of the
sub-bitstreams (tiled.py).  Frames of a GOP never interact (the batch index is part of every key:
/root/reference/shared/utils.py:10-42, sender/encoder/codec_pipeline.py:456-458), so tiles coded as frames of
one GOP, as one GOP per rank, or one by one give the same per-tile streams and the same reconstruction.

CPU (`-m "not gpu"`): the cutter, the bundle format, and a world-size-2 `gloo` run of compress_tiled /
decompress_tiled in which the codec is the CPU oracle (real containers through the real exchange path).
GPU (`-m gpu`): the same with the HIP codec — C5 at full size on one GPU against the oracle per tile, and a
2-rank gloo run in which both ranks share the GPU, checked against the single-process result.
"""
import os
import socket

import numpy as np
import pytest

from conftest import pkg

torch = pytest.importorskip("torch")

SETTINGS = [[1.0, 0.0], [0.0, 1.0], [1, 1]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _small_tiles(wl, n=5):
    return [wl.sphere_shell(20 + 2 * t, 7.0 + t, seed=40 + t, offset=(64 * t - 128, -32 * t, 7 * t)) for t in range(n)]


def _same_frames(a, b):
    return len(a) == len(b) and all(np.array_equal(x["points"], y["points"]) and np.array_equal(x["colors"], y["colors"])
                                    for x, y in zip(a, b))


# ----------------------------------------------------------------------------------- CPU
def test_cut_tiles_partitions_the_scan(wl):
    tiled = pkg("tiled")
    rng = np.random.default_rng(5)
    pts = np.unique(rng.integers(-300, 300, (20000, 3)), axis=0).astype(np.int16)
    frame = {"points": pts, "colors": rng.random((pts.shape[0], 3))}
    block = tiled.octree_blocks(pts, 8)
    assert all(b % tiled.ALIGN == 0 and b & (b - 1) == 0 for b in block)
    tiles, origins = tiled.cut_tiles(frame, block)
    assert len(tiles) >= 8 and len(tiles) == len(origins) == len(set(origins))
    assert origins == sorted(origins)                                  # lexicographic block order
    for t, o in zip(tiles, origins):
        p = t["points"].astype(np.int64)
        assert t["points"].dtype == np.int16 and t["colors"].dtype == np.float64
        assert ((p >= np.asarray(o)) & (p < np.asarray(o) + np.asarray(block))).all()
        assert all(v % b == 0 for v, b in zip(o, block))
    merged = tiled.merge_tiles(tiles)
    assert merged["points"].shape == pts.shape
    # the same rows with the same colours, only re-ordered
    def rows(f):
        k = np.lexsort(f["points"].T[::-1])
        return f["points"][k], f["colors"][k]
    a, b = rows(frame), rows(merged)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # a block that is not on the stride-32 lattice is refused: tiles must not share a latent voxel
    with pytest.raises(ValueError):
        tiled.cut_tiles(frame, (100, 64, 64))
    # one tile when the block holds everything
    one, _ = tiled.cut_tiles(frame, (2048, 2048, 2048))
    assert len(one) <= 8 and sum(t["points"].shape[0] for t in one) == pts.shape[0]


def test_fused_scan_cuts_into_its_eight_rooms(wl):
    tiled = pkg("tiled")
    scan = wl.fused_scan(40_000, seed=3, extent=(128, 128, 64))
    block = tiled.octree_blocks(scan["points"], 8)
    # the scan is centred on the origin, so the aligned grid of the bounding box's own power-of-two sides already
    # separates the eight rooms
    assert block == (256, 256, 128)
    tiles, origins = tiled.cut_tiles(scan, block)
    assert len(tiles) == 8 and [t["points"].shape[0] for t in tiles] == [5000] * 8
    assert origins[0] == (-256, -256, -128) and origins[-1] == (0, 0, 0)
    assert len(tiled.cut_tiles(scan, (128, 128, 64))[0]) == 8          # the rooms' own size cuts the same way
    off = {"points": scan["points"] + np.array([128, 0, 0], dtype=np.int16), "colors": scan["colors"]}
    assert tiled.octree_blocks(off["points"], 8) == (128, 256, 128)     # off-centre: x has to be halved once more


def test_substream_bundle_roundtrip():
    tiled = pkg("tiled")
    blobs = [b"", b"abc", bytes(1000)]
    assert tiled.unpack_substreams(tiled.pack_substreams(blobs)) == blobs
    with pytest.raises(ValueError):
        tiled.unpack_substreams(tiled.pack_substreams(blobs)[:-1])
    with pytest.raises(ValueError):
        tiled.unpack_substreams(b"XXXX\x00\x00\x00\x00")
    with pytest.raises(ValueError):
        tiled.unpack_substreams(b"PCCT\x7f\xff\xff\xff")


def test_decoded_frame_pack_roundtrip():
    tiled = pkg("tiled")
    rng = np.random.default_rng(1)
    frames = [{"points": rng.integers(-9, 9, (n, 3)).astype(np.int32), "colors": rng.random((n, 3)).astype(np.float32)}
              for n in (0, 1, 77)]
    assert _same_frames(tiled._unpack_frames(tiled._pack_frames(frames)), frames)


def _oracle_fns():
    from oracle.codec_ref import Oracle
    o = Oracle(threads=2)

    def compress(gop):
        frames = gop.pop("frames")
        return o.compress(frames, SETTINGS)[0], gop

    def decompress(data):
        return o.decompress(data), {}

    return compress, decompress


def _worker_oracle(rank, world, port, n_tiles, ret):
    import torch.distributed as dist
    tiled, wl = pkg("tiled"), pkg("workloads")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        compress, decompress = _oracle_fns()
        tiles = _small_tiles(wl, n_tiles)
        cpu = torch.device("cpu")
        mine = [dict(tiles[t]) for t in tiled.tiles_of_rank(n_tiles, rank, world)]
        bundles, _ = tiled.compress_tiled(compress, mine, [1, 3], cpu)
        # every rank holds every rank's containers; they equal what one process produces for the same deal
        ref, _ = tiled.compress_tiled_local(compress, [dict(t) for t in tiles], [1, 3], world)
        assert bundles == ref, (rank, [[len(c) for c in b] for b in bundles])
        # decode: own share, then the all-gather of the decoded tiles; tile order restored on every rank
        rec = tiled.decompress_tiled(decompress, bundles, n_tiles, cpu, which=1)
        one_by_one = [decompress(compress({"frames": [dict(t)]})[0][3])[0][0] for t in tiles]
        assert _same_frames(rec, one_by_one)
        assert tiled.merge_tiles(rec)["points"].shape[0] == sum(t["points"].shape[0] for t in tiles)
        # empty and large payloads through the byte all-gather
        got = tiled.all_gather_bytes(b"" if rank == 0 else bytes(range(256)) * 4097, cpu)
        assert got[0] == b"" and got[1] == bytes(range(256)) * 4097
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_tiles", [5, 2, 1])
def test_tiled_codec_gloo_world2_oracle_codec(n_tiles):
    """world size 2 on CPU; n_tiles = 1 leaves rank 1 without a tile (empty bundle through the exchange)"""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_oracle, args=(world, _free_port(), n_tiles, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_tiles_in_one_gop_equal_tiles_coded_alone_oracle(wl):
    """the independence the design rests on, on the oracle: a tile's decoded frame does not depend on its GOP"""
    tiled = pkg("tiled")
    compress, decompress = _oracle_fns()
    tiles = _small_tiles(wl, 3)
    b1, _ = tiled.compress_tiled_local(compress, [dict(t) for t in tiles], 3, 1)
    b3, _ = tiled.compress_tiled_local(compress, [dict(t) for t in tiles], 3, 3)
    r1 = tiled.decompress_tiled(decompress, b1, 3, torch.device("cpu"))
    r3 = tiled.decompress_tiled(decompress, b3, 3, torch.device("cpu"))
    assert _same_frames(r1, r3)


# ----------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def codec():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return pkg("codec_pipeline").CompressionPipeline(SETTINGS, slots=1), \
        pkg("codec_parallel").DecompressionPipeline(slots=1)


@pytest.mark.gpu
def test_c5_tiled_scan_full_size_equals_oracle(codec, oracle, wl):
    """workloads.tiled_scan(): 8 tiles x 500k voxels = BASELINE.json configs[4] on ONE GPU, (a) all tiles as the
    frames of one GOP and (b) as 8 single-tile GOPs (what 8 ranks hold after the all-gather): container bytes of
    every quality and every decoded tile equal the oracle's"""
    tiled = pkg("tiled")
    enc, dec = codec
    tiles = wl.tiled_scan()
    assert len(tiles) == 8 and all(t["points"].shape[0] == 500_000 for t in tiles)
    dev = torch.device("cuda", 0)
    # (a) one GOP of 8 frames
    ref, _ = oracle.compress([dict(t) for t in tiles], SETTINGS)
    b1, _ = tiled.compress_tiled_local(enc.compress, [dict(t) for t in tiles], [1, 2, 3], 1)
    for i, q in enumerate((1, 2, 3)):
        assert b1[0][i] == ref[q], f"8-tile GOP, container {q} differs from the oracle"
    rec = tiled.decompress_tiled(dec.decompress, b1, 8, dev, which=2)
    oref = oracle.decompress(ref[3])
    assert _same_frames(rec, oref)
    # (b) one GOP per tile
    b8, _ = tiled.compress_tiled_local(enc.compress, [dict(t) for t in tiles], [1, 2, 3], 8)
    for t in range(8):
        r1, _ = oracle.compress([dict(tiles[t])], SETTINGS)
        assert b8[t] == [r1[1], r1[2], r1[3]], f"tile {t} coded alone differs from the oracle"
    rec8 = tiled.decompress_tiled(dec.decompress, b8, 8, dev, which=2)
    # and the tile's reconstruction does not depend on the GOP it travelled in
    assert _same_frames(rec8, rec)
    scan = tiled.merge_tiles(rec8)
    assert scan["points"].shape[0] == 4_000_000


@pytest.mark.gpu
def test_c5_fused_scan_cut_code_merge(codec, wl):
    """one 4M-point frame -> cut_tiles -> 8 octree blocks -> coded as 2 shares -> decoded, merged: the scan comes back
    with its voxel count, without duplicates, every decoded voxel inside the block of its tile"""
    tiled = pkg("tiled")
    enc, dec = codec
    scan = wl.fused_scan(4_000_000, seed=0)
    block = tiled.octree_blocks(scan["points"], 8)
    tiles, origins = tiled.cut_tiles(scan, block)
    assert len(tiles) == 8 and block == (1024, 1024, 512)
    bundles, sides = tiled.compress_tiled_local(enc.compress, [dict(t) for t in tiles], 3, 2)
    assert sum(s["gop_info"]["num_points"] for s in sides) == 4_000_000
    rec = tiled.decompress_tiled(dec.decompress, bundles, 8, torch.device("cuda", 0))
    for r, t, o in zip(rec, tiles, origins):
        p = r["points"].astype(np.int64)
        assert p.shape[0] == t["points"].shape[0]
        assert ((p >= np.asarray(o)) & (p < np.asarray(o) + np.asarray(block))).all()
    merged = tiled.merge_tiles(rec)
    p = merged["points"].astype(np.int64) + 2048
    assert np.unique((p[:, 0] << 24) | (p[:, 1] << 12) | p[:, 2]).shape[0] == 4_000_000


def _worker_gpu(rank, world, port, n_tiles, ret):
    import torch.distributed as dist
    tiled, wl = pkg("tiled"), pkg("workloads")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        enc = pkg("codec_pipeline").CompressionPipeline(SETTINGS, slots=1)
        dec = pkg("codec_parallel").DecompressionPipeline(slots=1)
        tiles = [wl.room(60_000, seed=70 + t, extent=(128, 128, 64), offset=(-128 * (t & 1), -128 * ((t >> 1) & 1), 0))
                 for t in range(n_tiles)]
        cpu = torch.device("cpu")
        mine = [dict(tiles[t]) for t in tiled.tiles_of_rank(n_tiles, rank, world)]
        bundles, _ = tiled.compress_tiled(enc.compress, mine, [1, 3], cpu)
        ref, _ = tiled.compress_tiled_local(enc.compress, [dict(t) for t in tiles], [1, 3], world)
        assert bundles == ref, (rank, [[len(c) for c in b] for b in bundles])
        rec = tiled.decompress_tiled(dec.decompress, bundles, n_tiles, cpu, which=1)
        one = [dec.decompress(enc.compress({"frames": [dict(t)], "timestamps": {}})[0][3])[0][0] for t in tiles]
        assert _same_frames(rec, one)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_tiled_codec_gloo_world2_hip_codec():
    """two ranks (gloo) sharing the one GPU of the box, the real codec on both: what the ranks gather equals what a
    single process produces, and the gathered reconstruction equals every tile coded and decoded alone"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_gpu, args=(world, _free_port(), 3, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
