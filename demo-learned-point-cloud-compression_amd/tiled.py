"""Spatially tiled frames across the GPUs of one node (BASELINE.json config 5).

A 4M-point scan is cut into T octree blocks; every block is coded as an
independent frame (no halos — SURVEY.md §8e), tiles are dealt round-robin to
the ranks, each rank runs the whole codec for its tiles, and the only exchange
step is one variable-length all-gather of the per-rank sub-bitstreams over
RCCL/xGMI (`nccl` backend) — `gloo` on CPU in the tests.  The payloads are tens
of KB to a few MB, so the collective is latency-bound; it is done as one
all_gather of the lengths and one all_gather of max-length-padded uint8 rows.
"""
import struct

import numpy as np
import torch
import torch.distributed as dist

MAGIC = b"PCCT"


def tiles_of_rank(n_tiles, rank, world):
    return list(range(rank, n_tiles, world))


def pack_substreams(blobs):
    """list of byte strings -> one byte string (count, lengths, payloads)"""
    head = struct.pack(">4si", MAGIC, len(blobs)) + b"".join(struct.pack(">i", len(b)) for b in blobs)
    return head + b"".join(blobs)


def unpack_substreams(data):
    magic, n = struct.unpack_from(">4si", data, 0)
    if magic != MAGIC or n < 0:
        raise ValueError("not a tiled sub-bitstream bundle")
    lens = struct.unpack_from(">" + "i" * n, data, 8)
    pos = 8 + 4 * n
    out = []
    for ln in lens:
        if ln < 0 or pos + ln > len(data):
            raise ValueError("truncated tiled sub-bitstream bundle")
        out.append(bytes(data[pos:pos + ln]))
        pos += ln
    return out


def all_gather_bytes(local, device, group=None):
    """every rank contributes one byte string; returns the list of all ranks' strings
    (identical on every rank).  Two collectives: lengths, then padded payloads."""
    world = dist.get_world_size(group)
    n = torch.tensor([len(local)], dtype=torch.int64, device=device)
    lens = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(lens, n, group=group)
    lens = [int(v.item()) for v in lens]
    width = max(max(lens), 1)
    buf = torch.zeros(width, dtype=torch.uint8, device=device)
    if len(local):
        buf[:len(local)] = torch.from_numpy(np.frombuffer(local, dtype=np.uint8).copy()).to(device)
    rows = [torch.empty(width, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(rows, buf, group=group)
    return [bytes(rows[r][:lens[r]].cpu().numpy().tobytes()) for r in range(world)]


def compress_tiled(compress_fn, tiles, quality, device, group=None):
    """tiles: the frames (one per tile) of THIS rank's share, in tile order.
    compress_fn(gop) -> (compressed_data, sideinfo) is CompressionPipeline.compress.
    Returns (bundles, sideinfo): bundles[r] = list of containers produced by rank r
    (one container per rank, holding that rank's tiles as frames)."""
    gop = {"frames": tiles, "timestamps": {}}
    if tiles:
        out, side = compress_fn(gop)
        local = pack_substreams([out[quality]])
    else:
        side, local = {}, pack_substreams([])
    gathered = all_gather_bytes(local, device, group)
    return [unpack_substreams(g) for g in gathered], side


def assemble_tiles(bundles_decoded, n_tiles, world):
    """undo the round-robin deal: bundles_decoded[r] = list of decoded frames of rank r"""
    out = [None] * n_tiles
    for r in range(world):
        for j, t in enumerate(tiles_of_rank(n_tiles, r, world)):
            out[t] = bundles_decoded[r][j]
    return out
