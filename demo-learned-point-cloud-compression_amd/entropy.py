"""Entropy models with the CompressAI method names the reference calls
(SURVEY.md §8b): `entropy_bottleneck.compress/decompress`,
`gaussian_conditional.build_indexes/compress/decompress/lower_bound_scale`.

Symbol and index formation runs on the GPU (csrc/entropy.hip); the serial
rANS stream itself is coded on the host by the C++ coder in libpcc_hip.so
(csrc/rans_host.cpp), bit-compatible with CompressAI's format.  Tensors keep
CompressAI's [B, C, N] layout at this surface.
"""
import numpy as np
import torch

from . import runtime as _rt


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


class _Coder:
    def __init__(self, cdf, length, offset):
        self.cdf = np.ascontiguousarray(cdf, dtype=np.int32)
        self.length = np.ascontiguousarray(length, dtype=np.int32)
        self.offset = np.ascontiguousarray(offset, dtype=np.int32)
        self._dev = None

    @property
    def dev(self):
        """the same tables in HBM, for the GPU coder of container version 1 (made on first use)"""
        if self._dev is None:
            self._dev = _rt.RansDev(self.cdf, self.length, self.offset)
        return self._dev

    def encode(self, sym, idx):
        """sym, idx: int32 numpy [S, n] -> list of S byte strings"""
        return _rt.rans_encode_multi(np.ascontiguousarray(sym), np.ascontiguousarray(idx), self.cdf, self.length,
                                     self.offset)

    def decode(self, data, idx, out=None):
        idx = np.ascontiguousarray(idx)
        if idx.dtype != np.uint8:
            idx = idx.astype(np.int32, copy=False)
        return _rt.rans_decode(data, idx, self.cdf, self.length, self.offset, out=out)


class EntropyBottleneck:
    """factorized prior over z: per-channel CDF, symbols = round(z - median)."""

    def __init__(self, tensors, params):
        self.coder = _Coder(tensors["entropy_bottleneck.quantized_cdf"], tensors["entropy_bottleneck.cdf_length"],
                            tensors["entropy_bottleneck.offset"])
        self.medians_host = tensors["entropy_bottleneck.medians"].astype(np.float32)
        self.medians = _dev(self.medians_host, params.dev["g_a.conv0.weight"].device)
        self.channels = int(self.medians_host.shape[0])

    def _indexes(self, n):
        return np.repeat(np.arange(self.channels, dtype=np.int32), n)

    # fused form used by the pipeline: z rows [N,C] in coding order
    def compress_rows(self, rt, z_rows, defer=False, version=0):
        """returns (strings, z_hat rows).  With defer=True `strings` is a zero-argument function
        doing the host rANS (to be called once the stream has passed this point): z_hat is formed
        on the device and does not depend on the byte string.  version=1: the string in the GPU coder's
        interleaved form (container version 1)."""
        sym, zhat = rt.factorized_quant(z_rows, self.medians)
        n = z_rows.shape[0]
        if version == 1:
            strings = self.coder.dev.encode(rt, sym.reshape(1, -1), None, max(n, 1))
            return ((lambda: strings) if defer else strings), zhat
        sym_h = rt.to_host_async(sym, "z_sym")

        def finish():
            return self.coder.encode(sym_h.reshape(1, -1), self._indexes(n).reshape(1, -1))

        if defer:
            return finish, zhat
        rt.sync()
        return finish(), zhat

    def decode_host(self, strings, n):
        """host half of decompress: rANS decode of the z string -> int32 [C*n] (no GPU involved)"""
        return self.coder.decode(strings[0], self._indexes(n))

    def decompress_rows(self, rt, strings, n, sym=None, version=0):
        if version == 1:
            sym_d = self.coder.dev.decode(rt, strings[0], self.channels * n, None, max(n, 1)).view(self.channels, n)
            return rt.factorized_dequant(sym_d, self.medians)
        if sym is None:
            sym = self.decode_host(strings, n)
        sym_d = rt.to_device(sym.reshape(self.channels, n), torch.int32)
        return rt.factorized_dequant(sym_d, self.medians)

    # CompressAI-shaped surface ([1,C,N] tensors)
    def compress(self, x):
        rt = _rt.current()
        assert x.dim() == 3 and x.shape[0] == 1
        rows = rt.to_device(x[0].t().contiguous(), torch.float32)
        strings, _ = self.compress_rows(rt, rows)
        return strings

    def decompress(self, strings, size):
        rt = _rt.current()
        n = int(size[0] if not isinstance(size[0], (list, tuple)) else size[0][0])
        zhat = self.decompress_rows(rt, strings, n)
        return zhat.t().contiguous().unsqueeze(0)


class GaussianConditional:
    """conditional Gaussian prior over y: CDF chosen per element by a scale index."""

    def __init__(self, tensors, params):
        self.coder = _Coder(tensors["gaussian_conditional.quantized_cdf"],
                            tensors["gaussian_conditional.cdf_length"], tensors["gaussian_conditional.offset"])
        self.scale_table_host = tensors["gaussian_conditional.scale_table"].astype(np.float32)
        self.scale_table = _dev(self.scale_table_host, params.dev["g_a.conv0.weight"].device)
        self.scale_bound = np.float32(self.scale_table_host[0])

    def lower_bound_scale(self, scales):
        return torch.clamp(scales, min=float(self.scale_bound))

    # CompressAI-shaped surface ([B, C, N] tensors), as the reference calls it --------------
    def build_indexes(self, scales):
        """codec_pipeline.py:425 / codec_parallel.py:398: int32 tensor of the same shape"""
        rt = _rt.current()
        return rt.build_indexes(rt.to_device(scales, torch.float32), self.scale_table)

    def compress(self, inputs, indexes, means=None):
        """codec_pipeline.py:426-430: one rANS string per batch item over the flattened [C, N]"""
        rt = _rt.current()
        inputs = rt.to_device(inputs, torch.float32)
        means = rt.to_device(means, torch.float32) if means is not None else None
        sym = rt.quantize_symbols(inputs, means)
        b = sym.shape[0]
        sym_h = sym.cpu().numpy().reshape(b, -1)
        idx_h = indexes.cpu().numpy().astype(np.int32, copy=False).reshape(b, -1)
        return self.coder.encode(sym_h, idx_h)

    def decompress(self, strings, indexes):
        """codec_parallel.py:400: no means -> the raw integer residuals as floats, shape of `indexes`"""
        idx_h = indexes.cpu().numpy().astype(np.int32, copy=False)
        b = idx_h.shape[0]
        out = np.stack([self.coder.decode(strings[i], idx_h[i].reshape(-1)) for i in range(b)], 0)
        rt = _rt.current()
        return rt.to_device(out.reshape(idx_h.shape).astype(np.float32))

    # fused forms used by the pipeline ------------------------------------
    def compress_rows(self, rt, y_rows, params_rows, scale_q, version=0):
        """y_rows [N,C], params_rows [N,2C], scale_q device [Q,C] -> Q byte strings (version=1: interleaved form)"""
        q = scale_q.shape[0]
        if version == 1:
            sym, idx = rt.gaussian_quant_dev(y_rows, params_rows, scale_q, self.scale_table)
            return self.coder.dev.encode(rt, sym.reshape(q, -1), idx.reshape(q, -1))
        sym, idx, flag = rt.gaussian_quant16(y_rows, params_rows, scale_q, self.scale_table)
        sym_h = rt.to_host_async(sym, "y_sym")
        idx_h = rt.to_host_async(idx, "y_idx")
        flag_h = rt.to_host_async(flag, "y_flag")
        rt.sync()
        if int(flag_h[0]) != 0:     # a symbol outside int16: take the generic int32 form
            sym, idx = rt.gaussian_quant(y_rows, params_rows, scale_q, self.scale_table)
            sym_h, idx_h = sym.cpu().numpy(), idx.cpu().numpy()
        return self.coder.encode(sym_h.reshape(q, -1), idx_h.reshape(q, -1))

    def decompress_rows(self, rt, string, params_rows, scale_1, off_a, off_b, version=0):
        """one quality: returns y_hat rows [N,C] (offset de-quantisation applied)"""
        n, c = params_rows.shape[0], params_rows.shape[1] // 2
        idx = rt.gaussian_indexes8(params_rows, scale_1, self.scale_table)
        if version == 1:
            sym_d = self.coder.dev.decode(rt, string, c * n, idx.reshape(-1), 1).view(c, n)
            return rt.gaussian_dequant(sym_d, params_rows, scale_1, float(self.scale_bound), float(off_a), float(off_b))
        idx_h = rt.to_host_async(idx, "y_idx")
        rt.sync()
        stage = rt.pinned("y_dec", 4 * c * n)[:4 * c * n].view(torch.int32)
        self.coder.decode(string, idx_h.reshape(-1), out=stage.numpy())
        sym_d = rt.from_pinned_async(stage).view(c, n)
        return rt.gaussian_dequant(sym_d, params_rows, scale_1, float(self.scale_bound), float(off_a),
                                   float(off_b))
