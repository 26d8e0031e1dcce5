"""Spatially tiled frames across the GPUs of one node (BASELINE.json config 5).

A 4M-point scan is cut into T octree blocks (`cut_tiles`); every block is coded as an
independent frame — the batch index is part of every coordinate key, so frames of a GOP
never interact (shared/utils.py:32-35, codec_pipeline.py:456-458; no halos, SURVEY.md
§8e) — tiles are dealt round-robin to the ranks, each rank runs the whole codec for its
tiles as one GOP, and the only exchange step is one variable-length all-gather of the
per-rank sub-bitstreams over RCCL/xGMI (`nccl` backend; `gloo` on CPU in the tests).
The payloads are tens of KB to a few MB, so the collective is latency-bound; it is done
as one all_gather of the lengths and one all_gather of max-length-padded uint8 rows.

Decode is the mirror image (`decompress_tiled`): every rank decodes the container of its
own share out of the gathered bundles, the decoded tiles are gathered to every rank the
same way, `assemble_tiles` undoes the round-robin deal and `merge_tiles` joins the tiles
into the scan.  Without a process group (world size 1) the same functions run all tiles
in this process and no collective is issued.
"""
import os
import struct

import numpy as np
import torch
import torch.distributed as dist

MAGIC = b"PCCT"
ALIGN = 32          # block boundaries sit on the stride-32 lattice: no latent (stride 8) or hyper-latent (stride 32)
                    # voxel is shared between two tiles, so the union of the decoded tiles has no duplicate voxel


# ------------------------------------------------------------------ the tile cutter
def octree_blocks(points, n_tiles=8):
    """per-axis block sides (powers of two >= ALIGN) of an octree-style split of the scan's bounding box: start from
    the smallest power-of-two box that holds the extent on every axis and halve the longest side until at least
    `n_tiles` blocks of the aligned grid are occupied (8 blocks of a 1024 x 1024 x 512 scan: 512 x 512 x 256)"""
    p = np.asarray(points).astype(np.int64)
    lo, hi = p.min(0), p.max(0)
    side = [max(ALIGN, 1 << int(np.ceil(np.log2(max(int(hi[a] - lo[a]) + 1, 1))))) for a in range(3)]
    while True:
        b = np.floor_divide(p, np.asarray(side))
        if np.unique(b, axis=0).shape[0] >= n_tiles or max(side) <= ALIGN:
            return tuple(int(s) for s in side)
        a = int(np.argmax(side))
        side[a] //= 2


def cut_tiles(frame, block):
    """frame {"points": int[N,3], "colors": float[N,3]} -> (tiles, origins): one frame per occupied block of the grid
    `block` = (bx, by, bz) (multiples of ALIGN, aligned at multiples of themselves), in lexicographic block order; the
    points keep their coordinates and their order inside a tile.  origins[t] = the block's low corner."""
    block = np.asarray(block, dtype=np.int64)
    if block.shape != (3,) or (block <= 0).any() or (block % ALIGN).any():
        raise ValueError(f"block sides must be positive multiples of {ALIGN}, got {block.tolist()}")
    pts = np.asarray(frame["points"])
    cols = np.asarray(frame["colors"])
    b = np.floor_divide(pts.astype(np.int64), block)
    key = ((b[:, 0] + (1 << 20)) << 42) | ((b[:, 1] + (1 << 20)) << 21) | (b[:, 2] + (1 << 20))
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    ends = np.concatenate([starts[1:], [ks.shape[0]]])
    tiles, origins = [], []
    for s, e in zip(starts, ends):
        rows = order[s:e]
        tiles.append({"points": np.ascontiguousarray(pts[rows]), "colors": np.ascontiguousarray(cols[rows])})
        origins.append(tuple(int(v) for v in b[rows[0]] * block))
    return tiles, origins


def merge_tiles(tiles):
    """decoded tiles (tile order) -> the scan: concatenation, tile after tile"""
    if not tiles:
        return {"points": np.zeros((0, 3), np.int32), "colors": np.zeros((0, 3), np.float32)}
    return {"points": np.concatenate([np.asarray(t["points"]) for t in tiles], 0),
            "colors": np.concatenate([np.asarray(t["colors"]) for t in tiles], 0)}


# ------------------------------------------------------------------ the deal and the bundles
def tiles_of_rank(n_tiles, rank, world):
    return list(range(rank, n_tiles, world))


def pack_substreams(blobs):
    """list of byte strings -> one byte string (count, lengths, payloads)"""
    head = struct.pack(">4si", MAGIC, len(blobs)) + b"".join(struct.pack(">i", len(b)) for b in blobs)
    return head + b"".join(blobs)


def unpack_substreams(data):
    if len(data) < 8:
        raise ValueError("not a tiled sub-bitstream bundle")
    magic, n = struct.unpack_from(">4si", data, 0)
    if magic != MAGIC or n < 0 or 8 + 4 * n > len(data):
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


def _world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def all_gather_bytes(local, device, group=None):
    """every rank contributes one byte string; returns the list of all ranks' strings
    (identical on every rank).  Two collectives: lengths, then padded payloads."""
    world = _world(group)
    # (PCC_TILED_FORCE_COLLECTIVE=1: a world of one rank still goes through the two collectives — how bench.py's
    # PCC_BENCH_TILED=1 rehearsal exercises them over RCCL on a one-GPU box)
    if world == 1 and not (os.environ.get("PCC_TILED_FORCE_COLLECTIVE") == "1" and dist.is_available() and dist.is_initialized()):
        return [bytes(local)]
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


# ------------------------------------------------------------------ encode
def compress_tiled(compress_fn, tiles, quality, device, group=None):
    """tiles: the frames (one per tile) of THIS rank's share, in tile order.
    compress_fn(gop) -> (compressed_data, sideinfo) is CompressionPipeline.compress.
    quality: one key of compressed_data, or a list of keys (all those containers travel).
    Returns (bundles, sideinfo): bundles[r] = list of containers produced by rank r
    (one container per requested quality, holding that rank's tiles as frames;
    empty for a rank without tiles)."""
    quals = list(quality) if isinstance(quality, (list, tuple)) else [quality]
    gop = {"frames": tiles, "timestamps": {}}
    if tiles:
        out, side = compress_fn(gop)
        local = pack_substreams([out[q] for q in quals])
    else:
        side, local = {}, pack_substreams([])
    gathered = all_gather_bytes(local, device, group)
    return [unpack_substreams(g) for g in gathered], side


def compress_tiled_local(compress_fn, tiles, quality, world):
    """the same deal inside ONE process (no collective): the n tiles are dealt round-robin to `world` shares, every
    share is coded as its own GOP; returns (bundles, [sideinfo per share]) shaped as compress_tiled returns them —
    what a world of that size would hold after its all-gather"""
    quals = list(quality) if isinstance(quality, (list, tuple)) else [quality]
    bundles, sides = [], []
    for r in range(world):
        share = [tiles[t] for t in tiles_of_rank(len(tiles), r, world)]
        if share:
            out, side = compress_fn({"frames": share, "timestamps": {}})
            bundles.append([out[q] for q in quals])
        else:
            side = {}
            bundles.append([])
        sides.append(side)
    return bundles, sides


# ------------------------------------------------------------------ decode
def _pack_frames(frames):
    parts = [struct.pack(">i", len(frames))]
    for f in frames:
        p = np.ascontiguousarray(f["points"], dtype=np.int32)
        c = np.ascontiguousarray(f["colors"], dtype=np.float32)
        parts += [struct.pack(">i", p.shape[0]), p.tobytes(), c.tobytes()]
    return b"".join(parts)


def _unpack_frames(data):
    n = struct.unpack_from(">i", data, 0)[0]
    pos, out = 4, []
    for _ in range(n):
        m = struct.unpack_from(">i", data, pos)[0]
        pos += 4
        p = np.frombuffer(data, dtype=np.int32, count=3 * m, offset=pos).reshape(m, 3).copy()
        pos += 12 * m
        c = np.frombuffer(data, dtype=np.float32, count=3 * m, offset=pos).reshape(m, 3).copy()
        pos += 12 * m
        out.append({"points": p, "colors": c})
    return out


def decompress_tiled(decompress_fn, bundles, n_tiles, device, which=0, group=None, gather=True):
    """bundles: what compress_tiled returned (bundles[r] = containers of rank r).  Every rank decodes container
    `which` of ITS OWN bundle with decompress_fn (DecompressionPipeline.decompress, numpy output); gather=True:
    the decoded tiles of all ranks are exchanged (the same variable-length all-gather) and the list of the
    n_tiles decoded tiles in tile order is returned on every rank; gather=False: only this rank's tiles
    (list in the order of tiles_of_rank).  World size 1: all bundles are decoded here."""
    world, rank = _world(group), _rank(group)
    if world == 1:
        # one process holds every rank's bundle: decode them all
        decoded = [decompress_fn(b[which])[0] if b else [] for b in bundles]
        return assemble_tiles(decoded, n_tiles, len(bundles))
    if len(bundles) != world:
        raise ValueError(f"{len(bundles)} bundles for a world of {world}")
    mine = decompress_fn(bundles[rank][which])[0] if bundles[rank] else []
    if not gather:
        return mine
    gathered = all_gather_bytes(_pack_frames(mine), device, group)
    return assemble_tiles([_unpack_frames(g) for g in gathered], n_tiles, world)


def assemble_tiles(bundles_decoded, n_tiles, world):
    """undo the round-robin deal: bundles_decoded[r] = list of decoded frames of rank r"""
    out = [None] * n_tiles
    for r in range(world):
        share = tiles_of_rank(n_tiles, r, world)
        if len(bundles_decoded[r]) != len(share):
            raise ValueError(f"rank {r} decoded {len(bundles_decoded[r])} tiles, its share is {len(share)}")
        for j, t in enumerate(share):
            out[t] = bundles_decoded[r][j]
    return out
