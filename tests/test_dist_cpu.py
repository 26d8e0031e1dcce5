"""world_size-2 `gloo` test of the only exchange step on the path: the variable-length
all-gather of per-rank sub-bitstreams for spatially tiled frames (tiled.py, BASELINE.json
config 5).  The codec itself is replaced by a deterministic byte generator here (no GPU);
the GPU codec is covered by the `-m gpu` tests."""
import hashlib
import os
import socket

import pytest

from conftest import pkg

torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_compress(gop):
    """stands in for CompressionPipeline.compress: container bytes depend on the tiles"""
    frames = gop.pop("frames")
    blob = b"".join(hashlib.sha256(repr(f).encode()).digest() * (1 + (len(repr(f)) % 5)) for f in frames)
    return {0: frames, 1: b"Q1" + blob, 2: b"Q2" + blob[::-1]}, gop


def _worker(rank, world, port, n_tiles, ret):
    import torch.distributed as dist
    tiled = pkg("tiled")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = tiled.tiles_of_rank(n_tiles, rank, world)
        tiles = [{"tile": t, "payload": "x" * (3 * t + 1)} for t in mine]
        bundles, _ = tiled.compress_tiled(_fake_compress, tiles, 2, torch.device("cpu"))
        # every rank sees every rank's container(s), identical everywhere
        expect = []
        for r in range(world):
            ts = [{"tile": t, "payload": "x" * (3 * t + 1)} for t in tiled.tiles_of_rank(n_tiles, r, world)]
            expect.append([_fake_compress({"frames": ts})[0][2]] if ts else [])
        assert bundles == expect, (rank, [len(b) for b in bundles])
        # round-robin deal is undone by assemble_tiles
        decoded = [[("dec", t) for t in tiled.tiles_of_rank(n_tiles, r, world)] for r in range(world)]
        assert tiled.assemble_tiles(decoded, n_tiles, world) == [("dec", t) for t in range(n_tiles)]
        # empty and large payloads through the byte all-gather
        got = tiled.all_gather_bytes(b"" if rank == 0 else bytes(range(256)) * 4097, torch.device("cpu"))
        assert got[0] == b"" and got[1] == bytes(range(256)) * 4097
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_tiles", [8, 3, 1])
def test_tiled_allgather_gloo_world2(n_tiles):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_tiles, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_substream_bundle_roundtrip():
    tiled = pkg("tiled")
    blobs = [b"", b"abc", bytes(1000)]
    assert tiled.unpack_substreams(tiled.pack_substreams(blobs)) == blobs
    with pytest.raises(ValueError):
        tiled.unpack_substreams(tiled.pack_substreams(blobs)[:-1])
    with pytest.raises(ValueError):
        tiled.unpack_substreams(b"XXXX\x00\x00\x00\x00")
