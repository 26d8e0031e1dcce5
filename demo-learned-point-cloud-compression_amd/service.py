"""The callers and data formats on either side of the codec path,
restated without sockets so that the MI355X codec can be dropped into the demo's services and
exercised end to end:

  sample / compress_batch / serialize_data   sender/encoder/encoder.py:82-153
  segment_items                               sender/media_server/media_server.py:119-141
  decode_segment                              receiver/decoder/decoder.py:50-77
  pack_playout_frame                          receiver/client/client.py:139-146
  make_bitstream / read_bitstream             the older per-frame container of codec_single.py
                                              (sender/encoder/codec_pipeline.py:520-588,
                                               receiver/decoder/codec_parallel.py:218-264)

ZeroMQ sockets, the HTTP server, the MPD and the GUI stay in the reference (out of scope).
"""
import pickle
import struct
import time

import numpy as np


# ------------------------------------------------------------------ sender side
def sample(batch, segment_duration, target_fps):
    """The GOP the encoder service hands to compress() (sender/encoder/encoder.py:95-129): of the frames captured during
    one segment, the n = segment_duration * target_fps whose capture times lie nearest to n equally spaced instants from
    the first frame's on (the earliest frame wins a tie; one frame may serve two instants).  The chosen frames lose their
    "timestamp" key, which moves into the GOP's timestamps — as the reference does it."""
    stamps = np.asarray([frame["timestamp"] for frame in batch], dtype=np.float64)
    n = int(segment_duration * target_fps)
    instants = stamps[0] + np.arange(n, dtype=np.float64) * (segment_duration / n)
    nearest = np.abs(stamps[None, :] - instants[:, None]).argmin(axis=1)       # first minimum = the reference's min()
    frames = [batch[i] for i in nearest]
    return {"frames": frames, "timestamps": {"capturing": [f.pop("timestamp") for f in frames], "sampling": time.time()}}


def compress_batch(codec, gop, segment_duration, target_fps):
    """encoder.py:131-145: the dict the encoder service pickles and pushes to the media server"""
    gop["segment_duration"] = segment_duration
    gop["frame_rate"] = target_fps
    compressed_data, sideinfo = codec.compress(gop)
    return {"compressed_data": compressed_data, "sideinfo": sideinfo}


def serialize_data(data):
    return pickle.dumps(data)


def segment_items(segment, publish_offset=3.0, segment_duration=1.0):
    """What the media server writes per representation: segment number and
    pickle((payload, sideinfo)) (media_server.py:123-141).  Returns (segment_number,
    {representation id: bytes of the .bin file})."""
    sideinfo = segment["sideinfo"]
    data = segment["compressed_data"]
    cap = sideinfo["timestamps"]["capturing"]
    publishing = sum(cap) / len(cap) + publish_offset
    number = int(np.floor(publishing / segment_duration))
    return number, {key: pickle.dumps((data[key], sideinfo)) for key in sorted(data)}


# ------------------------------------------------------------------ receiver side
def client_handoff(segment_file, segment_number, quality, codec_info):
    """client.py:108-115: downloaded .bin file -> the dict the client pushes to the decoder"""
    data, sideinfo = pickle.loads(segment_file)
    sideinfo["ID"] = segment_number
    sideinfo["quality"] = quality
    sideinfo["codec_info"] = codec_info
    sideinfo["timestamps"]["client_received"] = time.time()
    return {"data": pickle.dumps(data), "sideinfo": sideinfo}


def decode_segment(codec, segment):
    """decoder.py:50-77 without the sockets: `segment` = {"data": pickled payload, "sideinfo": {...}}"""
    sideinfo = segment["sideinfo"]
    data = segment["data"]
    sideinfo["timestamps"]["decoder_received"] = time.time()
    if sideinfo.get("codec_info") == "unified":
        data, codec_info = codec.decompress(pickle.loads(data))
    else:
        data = pickle.loads(data)
        now = time.time()
        codec_info = {"time_measurements": {k: 0.0 for k in (
            "bitstream_reading", "geometry_decompression", "factorized_model", "hyper_synthesis",
            "guassian_model", "synthesis_transform", "postprocessing")},
            "timestamps": {"codec_start": now, "codec_end": now}}
    sideinfo["timestamps"].update(codec_info["timestamps"])
    sideinfo["time_measurements"] = codec_info["time_measurements"]
    sideinfo["timestamps"]["decoder_finished"] = time.time()
    return {"data": data, "sideinfo": sideinfo}


def pack_playout_frame(frame):
    """WebSocket frame for the visualizer: float32 xyz (points + 100) then uint8 rgb (255 * colours)
    (client.py:139-146, receiver/visualizer/main.js:46-60)"""
    points = np.asarray(frame["points"]) + 100
    colors = 255 * np.asarray(frame["colors"])
    return np.array(points, dtype=np.float32).tobytes() + np.array(colors, dtype=np.uint8).tobytes()


# ------------------------------------------------------------------ legacy per-frame container
def make_bitstream(y_strings, z_strings, y_shapes, z_shapes, points_streams, ks, q):
    """Older container, one header per frame (codec_pipeline.py:520-588, big-endian):
    int32 num_frames | f64 q_g | f64 q_a | F x ( int32 N_y | int32 N_z | int32 len_pts | int32 len_y |
    int32 len_z | int32 k1 | int32 k2 | int32 k3 | pts | y_string | z_string )"""
    num_frames = len(y_strings)
    parts = [struct.pack(">idd", num_frames, float(q[0]), float(q[1]))]
    for i in range(num_frames):
        points, y, z = points_streams[i], y_strings[i][0], z_strings[i][0]
        parts.append(struct.pack(">iiiiiiii", int(y_shapes[i]), int(z_shapes[i]), len(points), len(y), len(z),
                                 int(ks[0][i]), int(ks[1][i]), int(ks[2][i])))
        parts += [points, y, z]
    return b"".join(parts)


def read_bitstream(data):
    """inverse of make_bitstream (codec_parallel.py:218-264)"""
    buf = memoryview(data)
    pos = 0

    def take(fmt):
        nonlocal pos
        size = struct.calcsize(fmt)
        if pos + size > len(buf):
            raise ValueError("truncated per-frame container")
        v = struct.unpack_from(fmt, buf, pos)
        pos += size
        return v

    def take_bytes(n):
        nonlocal pos
        if n < 0 or pos + n > len(buf):
            raise ValueError("truncated per-frame container")
        b = bytes(buf[pos:pos + n])
        pos += n
        return b

    num_frames, q_g, q_a = take(">idd")
    y_strings, z_strings, y_shapes, z_shapes, points_streams, ks = [], [], [], [], [], [[], [], []]
    for _ in range(num_frames):
        n_y, n_z, lp, ly, lz, k1, k2, k3 = take(">iiiiiiii")
        ks[0].append(k1), ks[1].append(k2), ks[2].append(k3)
        points_streams.append(take_bytes(lp))
        y_strings.append(take_bytes(ly))
        z_strings.append(take_bytes(lz))
        y_shapes.append(n_y)
        z_shapes.append(n_z)
    return y_strings, z_strings, y_shapes, z_shapes, points_streams, ks, [q_g, q_a]
