"""DecompressionPipeline — drop-in for the reference's decoder operator
(receiver/decoder/codec_parallel.py:18-502; caller receiver/decoder/decoder.py:44,62).

`decompress(bytes) -> (list of {"points","colors"} per frame, sideinfo)` with
the reference's sideinfo keys (including the spelling "guassian_model") and
stage taxonomy D1..D6.  All stage work runs in libpcc_hip.so on the MI355X.
"""
import concurrent.futures
import os
import queue
import struct
import time

import numpy as np
import torch

from . import runtime as _rt
from . import utils
from .model import ColorModel, load_model_dir
from .native import NativeCodec
from .sparse import SparseTensor


class DecompressionPipeline:
    def __init__(self, device=0, slots=3, output="numpy", stage_sync=None, engine=None, base_path=None):
        self.device = torch.device("cuda", device)
        # "native": one pcc_decode_gop call per container (csrc/codec.hip); "ops": the reference's stage
        # methods one by one over the op-level C-ABI (same reconstruction, bit for bit)
        self.engine = engine or os.environ.get("PCC_ENGINE", "native")
        if self.engine not in ("native", "ops"):
            raise ValueError(f"engine must be 'native' or 'ops', got {self.engine!r}")
        # stage_sync=True: synchronise the stream at the end of every stage so that dec_time_measurements
        # holds per-stage wall times; False (default, PCC_STAGE_SYNC=1 overrides): stages are enqueued
        # back to back like the reference's asynchronous torch ops and only data hand-overs wait
        self.stage_sync = (os.environ.get("PCC_STAGE_SYNC", "0") == "1") if stage_sync is None else bool(stage_sync)
        # as the reference (codec_parallel.py:48-49); PCC_MODEL_BASE or the argument name another directory
        base_path = base_path or os.environ.get("PCC_MODEL_BASE", "./unified/results/")
        self.decompression_model = self.load_model(base_path)
        self.output = output                  # "numpy" (reference behaviour) or "device"
        self._slots = queue.Queue()
        if self.engine == "native":
            self.codecs = [NativeCodec(self.decompression_model.tensors, device) for _ in range(slots)]
            self.runtimes = [c.rt for c in self.codecs]
            for c in self.codecs:
                self._slots.put(c)
        else:
            self.runtimes = [_rt.Runtime(device) for _ in range(slots)]
            for r in self.runtimes:
                self._slots.put(r)
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(2, slots))

    def load_model(self, base_path):
        model_name = "demo_small"
        config, tensors = load_model_dir(base_path, model_name)
        decompression_model = ColorModel(config, tensors)
        decompression_model.load_state_dict(None)
        decompression_model.to(self.device)
        decompression_model.update()
        decompression_model.eval()
        return decompression_model

    # ------------------------------------------------------------------ main
    def decompress(self, compressed_data):
        """bitstream reading -> reconstruction (codec_parallel.py:141-171)"""
        if self.engine == "native":
            return self._decompress_native(compressed_data)
        t_start = time.time()
        rt = self._slots.get()
        try:
            with rt:
                y_strings, z_strings, y_shapes, z_shapes, points_streams, ks, q, t_1 = \
                    self.read_bitstream_batched(compressed_data)
                version = self.container_version(compressed_data)
                # the z string needs nothing from the GPU: decode it on the helper thread while the
                # geometry is decoded and the z coordinates are re-derived on the device (version 1: the
                # GPU coder decodes it in its place in the stream)
                eb = self.decompression_model.entropy_model.entropy_bottleneck
                z_sym = self._pool.submit(eb.decode_host, z_strings, int(z_shapes)) if version == 0 else None
                y_points, t_2 = self.geometry_decompression_step(points_streams)
                z_hats, t_3 = self.factorized_model_step_batched(z_strings, z_shapes, y_points, z_sym=z_sym,
                                                                 version=version)
                gaussian_params, t_4 = self.hyper_synthesis_step(z_hats)
                y_hat, t_5 = self.gaussian_model_step_batched(y_strings, y_shapes, y_points, q, gaussian_params,
                                                              version=version)
                reconstructed_pointcloud, t_6 = self.hyper_synthesis(y_hat, ks)
                final_data, t_7 = self.pack_batches(reconstructed_pointcloud, len(points_streams))
        finally:
            self._slots.put(rt)
        sideinfo = {"time_measurements": {}, "timestamps": {}}
        sideinfo["time_measurements"]["bitstream_reading"] = t_1
        sideinfo["time_measurements"]["geometry_decompression"] = t_2
        sideinfo["time_measurements"]["factorized_model"] = t_3
        sideinfo["time_measurements"]["hyper_synthesis"] = t_4
        sideinfo["time_measurements"]["guassian_model"] = t_5
        sideinfo["time_measurements"]["synthesis_transform"] = t_6
        sideinfo["timestamps"]["codec_start"] = t_start
        sideinfo["timestamps"]["codec_end"] = time.time()
        return final_data, sideinfo

    def _decompress_native(self, compressed_data):
        """the same call through pcc_decode_gop; packing as pack_batches (codec_parallel.py:474-502)"""
        t_start = time.time()
        codec = self._slots.get()
        try:
            with torch.cuda.stream(codec.stream):
                if self.output == "device":
                    coords, colors, offs, _, times = codec.decode(bytes(compressed_data))
                    batch = [{"points": coords[offs[i]:offs[i + 1], 1:], "colors": colors[offs[i]:offs[i + 1]]}
                             for i in range(len(offs) - 1)]
                else:
                    # NaN -> 0 and the colour clip of pack_batches run on the device; the host only slices
                    points, cols, offs, _, times = codec.decode(bytes(compressed_data), packed_host=True)
                    batch = [{"points": points[offs[i]:offs[i + 1]], "colors": cols[offs[i]:offs[i + 1]]}
                             for i in range(len(offs) - 1)]
        finally:
            self._slots.put(codec)
        sideinfo = {"time_measurements": times,
                    "timestamps": {"codec_start": t_start, "codec_end": time.time()}}
        return batch, sideinfo

    # ------------------------------------------------------------------ stages
    @staticmethod
    def container_version(compressed_data):
        """top byte of the first word: 0 = the reference's container, 1 = y / z strings in the GPU coder's form"""
        version = memoryview(compressed_data)[0] if len(compressed_data) else 0
        if version not in (0, 1):
            raise _rt.PccError(-5, "read_bitstream_batched", f"container version {version}")
        return int(version)

    def read_bitstream_batched(self, compressed_data):
        """Step 1: parse the container (codec_parallel.py:173-216)"""
        t0 = time.time()
        buf = memoryview(compressed_data)
        pos = 0

        def take(fmt):
            nonlocal pos
            size = struct.calcsize(fmt)
            if pos + size > len(buf):
                raise _rt.PccError(-5, "read_bitstream_batched", "truncated container")
            vals = struct.unpack_from(fmt, buf, pos)
            pos += size
            return vals

        def take_bytes(n):
            nonlocal pos
            if n < 0 or pos + n > len(buf):
                raise _rt.PccError(-5, "read_bitstream_batched", "truncated container")
            b = bytes(buf[pos:pos + n])
            pos += n
            return b

        num_frames, q_g, q_a = take(">idd")
        num_frames &= 0x00FFFFFF                  # the top byte is the container version
        q = [q_g, q_a]
        y_shapes, z_shapes, y_len, z_len = take(">iiii")
        y_strings = [take_bytes(y_len)]
        z_strings = [take_bytes(z_len)]
        points_streams, ks = [], [[], [], []]
        for _ in range(num_frames):
            p_len, k1, k2, k3 = take(">iiii")
            ks[0].append(k1)
            ks[1].append(k2)
            ks[2].append(k3)
            points_streams.append(take_bytes(p_len))
        return y_strings, z_strings, y_shapes, z_shapes, points_streams, ks, q, time.time() - t0

    def geometry_decompression_step(self, points_streams):
        """Step 2: latent coordinates of every frame (codec_parallel.py:266-289)"""
        t0 = time.time()
        y_points = [utils.gpcc_decode(s, 8) for s in points_streams]
        y_points = utils.stack_tensors(y_points)
        return y_points, time.time() - t0

    def factorized_model_step_batched(self, z_strings, z_shapes, y_points, z_sym=None, version=0):
        """Step 3: re-derive the z coordinates from the y coordinates with two
        stride-2 maps, decode z (codec_parallel.py:291-318)"""
        t0 = time.time()
        rt = _rt.current()
        n_frames = int(y_points[:, 0].max().item()) + 1 if y_points.shape[0] else 0
        latent_coordinates = SparseTensor(coordinates=y_points,
                                          features=torch.ones((y_points.shape[0], 1)),
                                          tensor_stride=8, device=self.device)
        latent_coordinates.cs.set_batches(n_frames)
        rt.y_latent = (y_points, latent_coordinates.cs)     # reused by the gaussian step of this call
        g_s = self.decompression_model.g_s
        latent_coordinates = g_s.down_conv(latent_coordinates)
        latent_coordinates = g_s.down_conv(latent_coordinates)
        z_view = utils.sort_coordset(latent_coordinates.cs)  # sort_points(latent_coordinates.C)
        z_points = z_view.C
        if z_points.shape[0] != int(z_shapes):
            raise _rt.PccError(-5, "factorized_model_step_batched",
                               f"container says N_z={int(z_shapes)}, coordinates give {z_points.shape[0]}")
        eb = self.decompression_model.entropy_model.entropy_bottleneck
        z_hat_rows = eb.decompress_rows(rt, z_strings, int(z_shapes),
                                        sym=z_sym.result() if z_sym is not None else None, version=version)
        z_hat = utils.sparse_from_rows(z_view, z_hat_rows)   # coordinates z_points, stride 32
        return z_hat, time.time() - t0

    def hyper_synthesis_step(self, z_hat):
        """Step 4: hyper synthesis"""
        t0 = time.time()
        gaussian_params = self.decompression_model.entropy_model.h_s(z_hat)
        if self.stage_sync:
            gaussian_params.rt.sync()
        return gaussian_params, time.time() - t0

    def gaussian_model_step_batched(self, y_strings, y_shapes, y_points, q, gaussian_params, version=0):
        """Step 5: decode y and de-quantise with offsets (codec_parallel.py:382-419)"""
        t0 = time.time()
        rt = _rt.current()
        em = self.decompression_model.entropy_model
        cached = getattr(rt, "y_latent", None)
        if cached is not None and cached[0] is y_points:
            y_cs = cached[1]                     # the coordinate set built from these points in step 3
        else:
            y_cs = SparseTensor(coordinates=y_points, features=torch.ones((y_points.shape[0], 1)),
                                tensor_stride=8, device=self.device).cs
        n_frames = y_cs.n_batch
        y_view = utils.sort_coordset(y_cs)       # sort_points(y_points)
        y_points = y_view.C
        if y_points.shape[0] != int(y_shapes):
            raise _rt.PccError(-5, "gaussian_model_step_batched",
                               f"container says N_y={int(y_shapes)}, geometry gives {y_points.shape[0]}")
        gaussian_params_feats = gaussian_params.features_at_coordinates(y_points)
        scale = em.scale_nn(np.asarray([q], dtype=np.float32)) + em.eps
        scale_dev = rt.to_device(np.ascontiguousarray(scale, dtype=np.float32))
        a, b = em.offsets_ab
        y_hat_rows = em.gaussian_conditional.decompress_rows(rt, y_strings[0], gaussian_params_feats, scale_dev,
                                                             a, b, version=version)
        y_hat = utils.sparse_from_rows(y_view, y_hat_rows)   # coordinates y_points, stride 8
        y_hat.cs.set_batches(n_frames)
        rt.y_latent = None
        return y_hat, time.time() - t0

    def hyper_synthesis(self, y_hat, ks):
        """Step 6: synthesis transform g_s with top-k pruning (named as in the
        reference, codec_parallel.py:465-472)"""
        t0 = time.time()
        reconstructed_pointcloud = self.decompression_model.g_s(y_hat, k=ks)
        reconstructed_pointcloud.rt.sync()
        return reconstructed_pointcloud, time.time() - t0

    def pack_batches(self, pointcloud, num_frames=None):
        """Step 7: split per frame, NaN -> 0, clip colours to [0,1]
        (codec_parallel.py:474-502)"""
        t0 = time.time()
        offs = pointcloud.cs.offsets
        if self.output == "device":
            coords, colors = pointcloud.C, pointcloud.F
            batch = [{"points": coords[offs[i]:offs[i + 1], 1:], "colors": colors[offs[i]:offs[i + 1]]}
                     for i in range(len(offs) - 1)]
            return batch, time.time() - t0
        points = pointcloud.C.cpu().numpy()
        colors = pointcloud.F.cpu().numpy()
        batch = []
        for i in range(len(offs) - 1):
            item_points = points[offs[i]:offs[i + 1], 1:]
            item_colors = np.nan_to_num(colors[offs[i]:offs[i + 1]], nan=0.0)
            item_colors = np.clip(item_colors * 255.0, 0, 255) / 255
            batch.append({"points": item_points, "colors": item_colors})
        return batch, time.time() - t0
