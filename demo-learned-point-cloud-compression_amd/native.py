"""Binding of the whole-GOP entry points of include/pcc.h (pcc_codec_create, pcc_encode_gop,
pcc_decode_gop): the native counterpart of CompressionPipeline.compress() / DecompressionPipeline.
decompress() (sender/encoder/codec_pipeline.py:196, receiver/decoder/codec_parallel.py:141).

The two pipeline classes of this package use it as their default engine; `engine="ops"` keeps the
op-by-op Python mirror of the reference's stage methods (same containers, byte for byte).
"""
import ctypes as C
import struct

import numpy as np
import torch

from . import _abi
from ._abi import check, PccError

ENC_STAGES = ("analysis", "hyper_analysis", "factorized_model", "hyper_synthesis", "geometry_compression",
              "gaussian_model", "bitstream_writing")
DEC_STAGES = ("bitstream_reading", "geometry_decompression", "factorized_model", "hyper_synthesis",
              "guassian_model", "synthesis_transform")


def pack_checkpoint(tensors):
    """dict name -> numpy array  ->  the "PCCW" blob pcc_codec_create reads (layout: include/pcc.h)"""
    parts = [b"PCCW", struct.pack("<I", len(tensors))]
    pos = 8

    def pad():
        nonlocal pos
        k = (-pos) % 8
        if k:
            parts.append(b"\0" * k)
            pos += k

    for name, arr in tensors.items():
        a = np.asarray(arr)
        if a.dtype.kind == "f":
            a, dt = np.ascontiguousarray(a, dtype="<f4"), 0
        elif a.dtype.kind in "iu":
            a, dt = np.ascontiguousarray(a, dtype="<i4"), 1
        else:
            raise TypeError(f"checkpoint tensor {name}: unsupported dtype {a.dtype}")
        nb = name.encode()
        head = struct.pack("<H", len(nb)) + nb + struct.pack("<BB", dt, a.ndim) + \
            struct.pack(f"<{a.ndim}I", *a.shape) + struct.pack("<Q", a.nbytes)
        parts.append(head)
        pos += len(head)
        pad()
        parts.append(a.tobytes())
        pos += a.nbytes
        pad()
    return b"".join(parts)


class NativeCodec:
    """one codec = weights in HBM + stream + device pool: one per in-flight call"""

    def __init__(self, tensors, device=0, container_version=0, seek_points=0):
        if not torch.cuda.is_available():
            raise RuntimeError("demo-learned-point-cloud-compression_amd needs a HIP device (no CPU fallback)")
        self.lib = _abi.lib()
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index)
        self.stream = torch.cuda.Stream(self.device)
        blob = pack_checkpoint(tensors)
        self.handle = self.lib.pcc_codec_create(blob, len(blob), self.device.index, C.c_void_p(self.stream.cuda_stream))
        if not self.handle:
            raise PccError(-1, "pcc_codec_create", self.lib.pcc_last_error().decode(errors="replace"))
        from .runtime import Runtime
        self.rt = Runtime.adopt(self.lib.pcc_codec_ctx(self.handle), self.stream, self.device)
        if container_version:
            check(self.lib.pcc_codec_set_container_version(self.handle, int(container_version)),
                  "pcc_codec_set_container_version")
        if seek_points:
            check(self.lib.pcc_codec_set_seek_points(self.handle, int(seek_points)), "pcc_codec_set_seek_points")

    def close(self):
        if getattr(self, "handle", None):
            self.rt.ctx = None
            self.lib.pcc_codec_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ encode
    def encode(self, coords, feats, n_frames, settings):
        """coords int32 [N,4] (b,x,y,z), feats float32 [N,4] = (1,r,g,b), both on this codec's device.
        Returns (list of Q containers, k[3][F], stage seconds dict)."""
        assert coords.is_cuda and feats.is_cuda and coords.dtype == torch.int32 and feats.dtype == torch.float32
        assert coords.is_contiguous() and feats.is_contiguous() and coords.shape[0] == feats.shape[0]
        n, nq = coords.shape[0], len(settings)
        q = (C.c_double * (2 * nq))(*[float(v) for s in settings for v in s[:2]])
        bufs = (_abi.PccBuf * nq)()
        k = (C.c_int64 * (3 * n_frames))()
        ts = (C.c_double * 7)()
        # inputs may have been produced on another stream: order this codec's stream after it
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        check(self.lib.pcc_encode_gop(self.handle, C.c_void_p(coords.data_ptr()), C.c_void_p(feats.data_ptr()), n,
                                      n_frames, q, nq, bufs, k, ts), "pcc_encode_gop")
        out = [C.string_at(bufs[i].data, bufs[i].len) for i in range(nq)]
        ks = [[int(k[s * n_frames + f]) for f in range(n_frames)] for s in range(3)]
        return out, ks, dict(zip(ENC_STAGES, ts))

    MAX_FRAMES = 32  # PCC_MAX_FRAMES_ARG of codec.hip

    def encode_frames(self, points, colors, settings):
        """the same on per-frame device tensors as the capture stage leaves them: points [n_f,3] int16 or int32,
        colors [n_f,3] float64 or float32 (one dtype per list).  Batch column, casts and the (1,r,g,b) rows are
        formed inside the library's kernels (pcc_encode_gop_frames); at most MAX_FRAMES frames."""
        nf, nq = len(points), len(settings)
        assert 1 <= nf <= self.MAX_FRAMES and len(colors) == nf
        pdt, cdt = points[0].dtype, colors[0].dtype
        assert pdt in (torch.int16, torch.int32) and cdt in (torch.float32, torch.float64)
        for p, c in zip(points, colors):
            assert p.is_cuda and c.is_cuda and p.dtype == pdt and c.dtype == cdt and p.is_contiguous()
            assert c.is_contiguous() and p.shape == c.shape and p.ndim == 2 and p.shape[1] == 3
        pp = (C.c_void_p * nf)(*[p.data_ptr() for p in points])
        cp = (C.c_void_p * nf)(*[c.data_ptr() for c in colors])
        ns = (C.c_int64 * nf)(*[int(p.shape[0]) for p in points])
        q = (C.c_double * (2 * nq))(*[float(v) for s in settings for v in s[:2]])
        bufs = (_abi.PccBuf * nq)()
        k = (C.c_int64 * (3 * nf))()
        ts = (C.c_double * 7)()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        check(self.lib.pcc_encode_gop_frames(self.handle, pp, 1 if pdt == torch.int16 else 0, cp,
                                             1 if cdt == torch.float64 else 0, ns, nf, q, nq, bufs, k, ts),
              "pcc_encode_gop_frames")
        out = [C.string_at(bufs[i].data, bufs[i].len) for i in range(nq)]
        ks = [[int(k[s * nf + f]) for f in range(nf)] for s in range(3)]
        return out, ks, dict(zip(ENC_STAGES, ts))

    def encode_host_frames(self, points, colors, settings):
        """the same on HOST arrays as compress(gop) receives them: numpy points [n_f,3] int16 or int32, colours [n_f,3]
        float64 or float32, C-contiguous, one dtype per list.  The library uploads them and overlaps the colours' PCIe
        leg with the key sort (pcc_encode_gop_host_frames)."""
        nf, nq = len(points), len(settings)
        assert 1 <= nf <= self.MAX_FRAMES and len(colors) == nf
        pdt, cdt = points[0].dtype, colors[0].dtype
        assert pdt in (np.int16, np.int32) and cdt in (np.float32, np.float64)
        for p, c in zip(points, colors):
            assert p.dtype == pdt and c.dtype == cdt and p.flags.c_contiguous and c.flags.c_contiguous
            assert p.shape == c.shape and p.ndim == 2 and p.shape[1] == 3
        pp = (C.c_void_p * nf)(*[p.ctypes.data for p in points])
        cp = (C.c_void_p * nf)(*[c.ctypes.data for c in colors])
        ns = (C.c_int64 * nf)(*[int(p.shape[0]) for p in points])
        q = (C.c_double * (2 * nq))(*[float(v) for s in settings for v in s[:2]])
        bufs = (_abi.PccBuf * nq)()
        k = (C.c_int64 * (3 * nf))()
        ts = (C.c_double * 7)()
        check(self.lib.pcc_encode_gop_host_frames(self.handle, pp, 1 if pdt == np.int16 else 0, cp,
                                                  1 if cdt == np.float64 else 0, ns, nf, q, nq, bufs, k, ts),
              "pcc_encode_gop_host_frames")
        out = [C.string_at(bufs[i].data, bufs[i].len) for i in range(nq)]
        ks = [[int(k[s * nf + f]) for f in range(nf)] for s in range(3)]
        return out, ks, dict(zip(ENC_STAGES, ts))

    # ------------------------------------------------------------------ decode
    MAX_ANNOUNCED = 1 << 26   # points: 1.5 GB of destination arrays

    def decode(self, data, packed_host=False):
        """container bytes -> (coords int32 [n,4] device, colors float32 [n,3] device, offsets, q, stage seconds).
        The two tensors are copies owned by the caller.  packed_host=True: the cloud as pack_batches returns it, in
        host memory — (points int32 [n,3] numpy, colours float32 [n,3] numpy clipped on the device, ...)."""
        info = _abi.PccCloudInfo()
        ts = (C.c_double * 6)()
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        if packed_host:
            # the arrays are sized by what the container announces (an upper bound) and handed to the decoder, which
            # queues the packing kernel and the transfers behind its last layer: one call, one synchronisation.  A
            # container that announces more than MAX_ANNOUNCED points is decoded first and its arrays sized afterwards.
            cap = C.c_int64(0)
            check(self.lib.pcc_container_points(buf, len(data), C.byref(cap), None), "pcc_container_points")
            if 0 < cap.value <= self.MAX_ANNOUNCED:
                pts = np.empty((cap.value, 3), dtype=np.int32)
                cols = np.empty((cap.value, 3), dtype=np.float32)
                check(self.lib.pcc_decode_gop_packed(self.handle, buf, len(data), C.c_void_p(pts.ctypes.data),
                                                     C.c_void_p(cols.ctypes.data), cap.value, C.byref(info), ts),
                      "pcc_decode_gop_packed")
                n = int(info.n_points)
                offsets = [int(info.h_offsets[i]) for i in range(info.n_offsets)]
                return pts[:n], cols[:n], offsets, [float(info.q_g), float(info.q_a)], dict(zip(DEC_STAGES, ts))
        check(self.lib.pcc_decode_gop(self.handle, buf, len(data), C.byref(info), ts), "pcc_decode_gop")
        n = int(info.n_points)
        offsets = [int(info.h_offsets[i]) for i in range(info.n_offsets)]
        if packed_host:
            pts = np.empty((n, 3), dtype=np.int32)
            cols = np.empty((n, 3), dtype=np.float32)
            if n:
                check(self.lib.pcc_decode_fetch_packed(self.handle, C.c_void_p(pts.ctypes.data),
                                                       C.c_void_p(cols.ctypes.data)), "pcc_decode_fetch_packed")
            return pts, cols, offsets, [float(info.q_g), float(info.q_a)], dict(zip(DEC_STAGES, ts))
        coords = torch.empty((n, 4), dtype=torch.int32, device=self.device)
        colors = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        if n:
            check(self.lib.pcc_decode_fetch(self.handle, C.c_void_p(coords.data_ptr()), C.c_void_p(colors.data_ptr())),
                  "pcc_decode_fetch")
        return coords, colors, offsets, [float(info.q_g), float(info.q_a)], dict(zip(DEC_STAGES, ts))
