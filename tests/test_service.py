"""the callers and data formats either side of the codec path
(service shells without sockets, the legacy per-frame container, the playout frame packing)."""
import pickle
import struct

import numpy as np
import pytest

from conftest import pkg


def _frames(wl, n):
    out = []
    for i in range(n):
        f = wl.sphere_shell(24 + 2 * (i % 3), 9.0 + (i % 3), seed=40 + i)
        f["timestamp"] = 100.0 + 0.1 * i
        out.append(f)
    return out


def test_sample_picks_nearest_frames(wl):
    """encoder.py:95-129: n = segment_duration*target_fps targets, nearest capture timestamp each"""
    service = pkg("service")
    batch = _frames(wl, 10)                                # captured at 10 fps for 1 s
    ids = [id(f) for f in batch]
    gop = service.sample(batch, 1.0, 5)
    assert len(gop["frames"]) == 5
    assert [ids.index(id(f)) for f in gop["frames"]] == [0, 2, 4, 6, 8]
    assert gop["timestamps"]["capturing"] == pytest.approx([100.0, 100.2, 100.4, 100.6, 100.8])
    assert all("timestamp" not in f for f in gop["frames"]) and "sampling" in gop["timestamps"]


def test_legacy_per_frame_container_kat():
    """byte layout authored from codec_pipeline.py:536-568 (big-endian, one header per frame)"""
    service = pkg("service")
    blob = service.make_bitstream([[b"yy1"], [b"y2"]], [[b"z"], [b"zz2"]], [7, 8], [1, 2], [b"PTS", b""],
                                  [[11, 12], [21, 22], [31, 32]], [1.0, 0.0])
    expect = (struct.pack(">idd", 2, 1.0, 0.0) +
              struct.pack(">iiiiiiii", 7, 1, 3, 3, 1, 11, 21, 31) + b"PTS" + b"yy1" + b"z" +
              struct.pack(">iiiiiiii", 8, 2, 0, 2, 3, 12, 22, 32) + b"" + b"y2" + b"zz2")
    assert blob == expect
    ys, zs, ny, nz, ps, ks, q = service.read_bitstream(blob)
    assert (ys, zs, ny, nz, ps, ks, q) == ([b"yy1", b"y2"], [b"z", b"zz2"], [7, 8], [1, 2], [b"PTS", b""],
                                           [[11, 12], [21, 22], [31, 32]], [1.0, 0.0])
    with pytest.raises(ValueError):
        service.read_bitstream(blob[:-2])


def test_playout_frame_packing():
    """client.py:139-146: float32 xyz (+100) then uint8 rgb (255 * colour, truncated)"""
    service = pkg("service")
    frame = {"points": np.array([[1, -2, 3], [40, 50, -60]], dtype=np.int32),
             "colors": np.array([[0.0, 0.5, 1.0], [0.999, 0.2, 0.25]], dtype=np.float32)}
    b = service.pack_playout_frame(frame)
    assert len(b) == 2 * 12 + 2 * 3
    assert np.array_equal(np.frombuffer(b[:24], dtype=np.float32).reshape(2, 3),
                          np.array([[101, 98, 103], [140, 150, 40]], dtype=np.float32))
    assert list(b[24:]) == [0, 127, 255, 254, 51, 63]


def test_segment_items_and_client_handoff():
    service = pkg("service")
    seg = {"compressed_data": {0: [{"points": 1}], 1: b"abc", 2: b"defg"},
           "sideinfo": {"timestamps": {"capturing": [10.0, 10.5], "sampling": 11.0}}}
    number, items = service.segment_items(seg, publish_offset=3.0, segment_duration=1.0)
    assert number == 13 and sorted(items) == [0, 1, 2]
    payload, side = pickle.loads(items[2])
    assert payload == b"defg" and side["timestamps"]["capturing"] == [10.0, 10.5]
    handed = service.client_handoff(items[2], number, 2, "unified")
    assert pickle.loads(handed["data"]) == b"defg"
    assert handed["sideinfo"]["ID"] == 13 and handed["sideinfo"]["codec_info"] == "unified"
    # raw representation (quality 0) passes through the decoder shell untouched
    raw = service.client_handoff(items[0], number, 0, "raw")
    out = service.decode_segment(None, raw)
    assert out["data"] == [{"points": 1}] and out["sideinfo"]["time_measurements"]["guassian_model"] == 0.0


@pytest.mark.gpu
def test_service_chain_end_to_end(wl):
    """capture frames -> sample -> compress_batch -> media-server items -> client hand-off ->
    decode_segment -> playout frames, with the MI355X codec in the middle"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    service = pkg("service")
    enc = pkg("codec_pipeline").CompressionPipeline([[1.0, 0.0], [0.0, 1.0], [1, 1]], slots=1)
    dec = pkg("codec_parallel").DecompressionPipeline(slots=1)
    batch = _frames(wl, 10)
    sizes = [batch[i]["points"].shape[0] for i in (0, 2, 4, 6, 8)]
    gop = service.sample(batch, 1.0, 5)
    seg = pickle.loads(service.serialize_data(service.compress_batch(enc, gop, 1.0, 5)))
    assert seg["sideinfo"]["segment_duration"] == 1.0 and seg["sideinfo"]["frame_rate"] == 5
    number, items = service.segment_items(seg)
    for quality in (1, 3):
        out = service.decode_segment(dec, service.client_handoff(items[quality], number, quality, "unified"))
        assert [f["points"].shape[0] for f in out["data"]] == sizes
        assert {"codec_start", "codec_end", "decoder_received", "decoder_finished", "capturing"} <= set(
            out["sideinfo"]["timestamps"])
        frames = [service.pack_playout_frame(f) for f in out["data"]]
        assert [len(b) for b in frames] == [15 * n for n in sizes]
