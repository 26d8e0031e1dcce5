"""Runtime: one pcc_ctx (device + HIP stream + scratch arena) with tensor-level
wrappers around the C-ABI.

torch is used here for what the task allows it for — device memory (caching
allocator), streams and host<->device copies.  Every computation on device
tensors goes through libpcc_hip.so; nothing here computes with torch ops.
One Runtime per in-flight compress()/decompress() call (the reference runs up
to three concurrent calls, sender/encoder/encoder.py:50).
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _abi
from ._abi import check, PccError

_tls = threading.local()

# Host thread pools must fit the CPU share of the box (see _abi.host_cpu_budget): the codec's
# host side is a handful of short tensor conversions, so a small intra-op pool is enough.
HOST_THREADS = int(os.environ.get("PCC_HOST_THREADS", "0")) or min(8, _abi.host_cpu_budget())
if torch.get_num_threads() > HOST_THREADS:
    torch.set_num_threads(HOST_THREADS)


def current():
    rt = getattr(_tls, "rt", None)
    if rt is None:
        raise RuntimeError("no active pcc Runtime on this thread (use `with runtime:`)")
    return rt


def _ptr(t):
    """device pointer of a tensor (NULL for None / empty).  A host tensor here would make a kernel
    dereference a host address — refuse it before it reaches the GPU."""
    if t is None or t.numel() == 0:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise TypeError("device tensor expected, got a host tensor")
    if not t.is_contiguous():
        raise TypeError("contiguous device tensor expected")
    return C.c_void_p(t.data_ptr())


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data)


class Runtime:
    def __init__(self, device=0, stream=None):
        if not torch.cuda.is_available():
            raise RuntimeError("pcc Runtime needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = _abi.lib()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        self.ctx = self.lib.pcc_create(device, C.c_void_p(self.stream.cuda_stream))
        if not self.ctx:
            raise PccError(-2, "pcc_create", self.lib.pcc_last_error().decode())
        self._stream_cm = None
        self._prev = None

    @classmethod
    def adopt(cls, ctx, stream, device):
        """Runtime view of a ctx owned by someone else (a pcc_codec): same helpers / profiler,
        close() leaves the ctx alone"""
        self = cls.__new__(cls)
        self.lib = _abi.lib()
        self.device = device
        self.stream = stream
        self.ctx = ctx
        self._owned = False
        self._stream_cm = None
        self._prev = None
        return self

    def close(self):
        if self.ctx and getattr(self, "_owned", True):
            self.lib.pcc_destroy(self.ctx)
        self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # `with rt:` makes rt the thread's active runtime and its stream torch's current stream
    def __enter__(self):
        self._prev = getattr(_tls, "rt", None)
        _tls.rt = self
        self._stream_cm = torch.cuda.stream(self.stream)
        self._stream_cm.__enter__()
        return self

    def __exit__(self, *exc):
        self._stream_cm.__exit__(*exc)
        _tls.rt = self._prev
        return False

    # ------------------------------------------------------------ helpers
    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def sync(self):
        check(self.lib.pcc_sync(self.ctx), "pcc_sync")

    def to_device(self, a, dtype=None):
        """host numpy / torch -> device tensor on this runtime's stream"""
        if isinstance(a, torch.Tensor):
            t = a
        else:
            t = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        if t.device != self.device:
            t = t.to(self.device, non_blocking=True)
        return t.contiguous()

    def timer_start(self):
        check(self.lib.pcc_timer_start(self.ctx), "pcc_timer_start")

    def timer_stop_ms(self):
        check(self.lib.pcc_timer_stop(self.ctx), "pcc_timer_stop")
        ms = C.c_float(0)
        check(self.lib.pcc_timer_elapsed_ms(self.ctx, C.byref(ms)), "pcc_timer_elapsed_ms")
        return ms.value

    # ------------------------------------------------------------ per-launch profiler
    def prof_enable(self, on=True, reserve=0, only=None, rows=-1):
        """reserve > 1 pre-creates that many event pairs (keeps event creation out of the timed region);
        only / rows: bracket just the entry points whose op name starts with `only` (and whose output row
        count is `rows`) — every event pair is a few microseconds of bubble on the stream"""
        check(self.lib.pcc_prof_only(self.ctx, only.encode() if only else None, int(rows)), "pcc_prof_only")
        check(self.lib.pcc_prof_enable(self.ctx, max(int(reserve), 1) if on else 0), "pcc_prof_enable")

    def prof_records(self):
        """[(op, ms, (d0,d1,d2,d3)), ...] of every C-ABI call since prof_enable (synchronises)"""
        out = []
        name = C.create_string_buffer(64)
        ms = C.c_float(0)
        dims = (C.c_int64 * 4)()
        for i in range(self.lib.pcc_prof_count(self.ctx)):
            check(self.lib.pcc_prof_get(self.ctx, i, name, 64, C.byref(ms), dims), "pcc_prof_get")
            out.append((name.value.decode(), float(ms.value), tuple(int(d) for d in dims)))
        return out

    def exclusive_scan(self, t, inplace=False):
        """exclusive prefix sum of an int32 / uint32 vector (mod 2^32): returns (scan, total tensor [1])"""
        n = t.shape[0]
        out = t if inplace else self.empty((n,), torch.int32)
        total = self.empty((1,), torch.int32)
        check(self.lib.pcc_exclusive_scan_u32(self.ctx, _ptr(t), _ptr(out), n, _ptr(total)), "pcc_exclusive_scan_u32")
        return out, total

    def count_nonneg(self, t):
        cnt = C.c_int64(0)
        check(self.lib.pcc_count_nonneg(self.ctx, _ptr(t), t.numel(), C.byref(cnt)), "pcc_count_nonneg")
        return cnt.value

    # ------------------------------------------------------------ keys / order
    def morton_keys(self, coords):
        n = coords.shape[0]
        keys = self.empty((n,), torch.int64)
        flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        check(self.lib.pcc_morton_keys(self.ctx, _ptr(coords), n, _ptr(keys), _ptr(flag)), "pcc_morton_keys")
        if n and int(flag.item()) != 0:
            raise PccError(_abi.PCC_E_RANGE, "pcc_morton_keys",
                           "coordinate outside [-32768,32767] or batch index outside [0,65534]")
        return keys

    def keys_to_coords(self, keys):
        n = keys.shape[0]
        coords = self.empty((n, 4), torch.int32)
        check(self.lib.pcc_keys_to_coords(self.ctx, _ptr(keys), n, _ptr(coords)), "pcc_keys_to_coords")
        return coords

    def linear_keys(self, coords):
        n = coords.shape[0]
        keys = self.empty((n,), torch.int64)
        check(self.lib.pcc_linear_keys(self.ctx, _ptr(coords), n, _ptr(keys)), "pcc_linear_keys")
        return keys

    def sort_pairs(self, keys, signed=False):
        """sorts `keys` in place, returns the permutation (int32 tensor holding uint32 indices)"""
        n = keys.shape[0]
        perm = self.empty((n,), torch.int32)
        check(self.lib.pcc_sort_pairs(self.ctx, _ptr(keys), _ptr(perm), n, 1 if signed else 0), "pcc_sort_pairs")
        return perm

    def sort_coords(self, coords):
        n = coords.shape[0]
        perm = self.empty((n,), torch.int32)
        check(self.lib.pcc_sort_coords(self.ctx, _ptr(coords), n, _ptr(perm)), "pcc_sort_coords")
        return perm

    def gather_rows(self, src, perm):
        n = perm.shape[0]
        row_elems = 1
        for d in src.shape[1:]:
            row_elems *= int(d)
        row_bytes = src.element_size() * row_elems
        dst = self.empty((n,) + tuple(src.shape[1:]), src.dtype)
        check(self.lib.pcc_gather_rows(self.ctx, _ptr(src), _ptr(perm), n, row_bytes, _ptr(dst)), "pcc_gather_rows")
        return dst

    def check_unique(self, sorted_keys):
        dup = C.c_int(0)
        check(self.lib.pcc_check_unique(self.ctx, _ptr(sorted_keys), sorted_keys.shape[0], C.byref(dup)),
              "pcc_check_unique")
        return dup.value == 0

    def batch_offsets(self, keys, n_batch):
        out = (C.c_int64 * (n_batch + 1))()
        check(self.lib.pcc_batch_offsets(self.ctx, _ptr(keys), keys.shape[0], n_batch, out), "pcc_batch_offsets")
        return [int(v) for v in out]

    # ------------------------------------------------------------ pyramid / maps
    def down_coords(self, keys, child_shift, m_known=None):
        """m_known: the parent count from level_counts — then the call does not synchronise"""
        n = keys.shape[0]
        pkeys = self.empty((n,), torch.int64)
        nbr8 = self.empty((8 * n,), torch.int32)
        parent_of = self.empty((n,), torch.int32)
        if m_known is None:
            m = C.c_int64(0)
            check(self.lib.pcc_down_coords(self.ctx, _ptr(keys), n, child_shift, _ptr(pkeys), _ptr(nbr8), n,
                                           _ptr(parent_of), C.byref(m)), "pcc_down_coords")
            m = m.value
        else:
            m = int(m_known)
            check(self.lib.pcc_down_coords_known(self.ctx, _ptr(keys), n, child_shift, _ptr(pkeys), _ptr(nbr8), n,
                                                 _ptr(parent_of), m), "pcc_down_coords_known")
        return pkeys[:m], nbr8[:8 * m].view(8, m), parent_of

    def level_counts(self, keys, child_shift, levels):
        """([rows of the `levels` successive parent sets of a sorted key set], duplicate-rows flag): one pass"""
        cnt = (C.c_int64 * levels)()
        dup = C.c_int(0)
        check(self.lib.pcc_level_counts(self.ctx, _ptr(keys), keys.shape[0], child_shift, levels, cnt, C.byref(dup)),
              "pcc_level_counts")
        return [int(v) for v in cnt], bool(dup.value)

    def derive_map_up(self, nbr_parent, n_parents, parent_rows=None, remap=None):
        nbr = self.empty((27, 8 * n_parents), torch.int32)
        check(self.lib.pcc_derive_map_up(self.ctx, _ptr(nbr_parent), nbr_parent.stride(0),
                                         _ptr(parent_rows) if parent_rows is not None else C.c_void_p(0),
                                         _ptr(remap) if remap is not None else C.c_void_p(0), n_parents,
                                         _ptr(nbr)), "pcc_derive_map_up")
        return nbr

    def subset_map_up(self, nbr_parent, keep, remap):
        """rule book [27, n_keep] of the rows `keep` of the 8 n generative children of a level with book nbr_parent"""
        n_keep = keep.shape[0]
        nbr = self.empty((27, n_keep), torch.int32)
        check(self.lib.pcc_subset_map_up(self.ctx, _ptr(nbr_parent), nbr_parent.stride(0), _ptr(keep), _ptr(remap),
                                         n_keep, _ptr(nbr)), "pcc_subset_map_up")
        return nbr

    def gather_map_columns(self, nbr, rows):
        """(rule book [K, m] of the subset rows `rows` (int32, -1 = absent) of a level with book nbr [K, n],
        self [m] = j where rows[j] >= 0 else -1)"""
        k_vol, m = nbr.shape[0], rows.shape[0]
        out = self.empty((k_vol, m), torch.int32)
        me = self.empty((m,), torch.int32)
        check(self.lib.pcc_gather_map_columns(self.ctx, _ptr(nbr), k_vol, nbr.stride(0), _ptr(rows), m, _ptr(out),
                                              _ptr(me)), "pcc_gather_map_columns")
        return out, me

    def derive_map_down(self, nbr_parent, nbr8, parent_of, keys, child_shift):
        n = keys.shape[0]
        nbr = self.empty((27, n), torch.int32)
        check(self.lib.pcc_derive_map_down(self.ctx, _ptr(nbr_parent), nbr_parent.shape[1], _ptr(nbr8),
                                           _ptr(parent_of), _ptr(keys), n, child_shift, _ptr(nbr)),
              "pcc_derive_map_down")
        return nbr

    def inverse_rows(self, rows, n):
        remap = self.empty((n,), torch.int32)
        check(self.lib.pcc_inverse_rows(self.ctx, _ptr(rows), rows.shape[0], n, _ptr(remap)), "pcc_inverse_rows")
        return remap

    def up_coords(self, keys, child_shift):
        n = keys.shape[0]
        ckeys = self.empty((8 * n,), torch.int64)
        check(self.lib.pcc_up_coords(self.ctx, _ptr(keys), n, child_shift, _ptr(ckeys)), "pcc_up_coords")
        return ckeys

    def up_coords_rows(self, keys, child_shift, rows):
        """Keys of the listed generative children (rows[i] = 8 p + o) without the 8 n keys being formed."""
        m = rows.shape[0]
        ckeys = self.empty((m,), torch.int64)
        check(self.lib.pcc_up_coords_rows(self.ctx, _ptr(keys), keys.shape[0], child_shift, _ptr(rows), m, _ptr(ckeys)),
              "pcc_up_coords_rows")
        return ckeys

    def build_map(self, keys, stride):
        n = keys.shape[0]
        nbr = self.empty((27, n), torch.int32)
        check(self.lib.pcc_build_map(self.ctx, _ptr(keys), n, stride, _ptr(nbr)), "pcc_build_map")
        return nbr

    def lookup(self, keys, qkeys):
        m = qkeys.shape[0]
        rows = self.empty((m,), torch.int32)
        check(self.lib.pcc_lookup(self.ctx, _ptr(keys), keys.shape[0], _ptr(qkeys), m, _ptr(rows)), "pcc_lookup")
        return rows

    def gather_rows_or_zero(self, src, rows):
        m, c = rows.shape[0], src.shape[1]
        dst = self.empty((m, c), torch.float32)
        check(self.lib.pcc_gather_rows_or_zero(self.ctx, _ptr(src), _ptr(rows), m, c, _ptr(dst)),
              "pcc_gather_rows_or_zero")
        return dst

    # ------------------------------------------------------------ layers
    def sparse_conv(self, x, nbr, w, b, relu):
        k_vol, n_out = nbr.shape
        cin, cout = w.shape[1], w.shape[2]
        assert x.shape[1] == cin and w.shape[0] == k_vol, (x.shape, w.shape, nbr.shape)
        out = self.empty((n_out, cout), torch.float32)
        check(self.lib.pcc_sparse_conv(self.ctx, _ptr(x), x.shape[0], _ptr(nbr), k_vol, nbr.stride(0), n_out,
                                       _ptr(w), _ptr(b), cin, cout, 1 if relu else 0, _ptr(out)),
              "pcc_sparse_conv")
        return out

    def sparse_conv_head(self, x, nbr, w, b, relu, head_w, head_b):
        """conv layer + fused 1x1 head (cout -> 1): returns (features [n,cout], logits [n])"""
        k_vol, n_out = nbr.shape
        cin, cout = w.shape[1], w.shape[2]
        assert x.shape[1] == cin and w.shape[0] == k_vol and head_w.numel() == cout
        out = self.empty((n_out, cout), torch.float32)
        logits = self.empty((n_out,), torch.float32)
        check(self.lib.pcc_sparse_conv_head(self.ctx, _ptr(x), x.shape[0], _ptr(nbr), k_vol, nbr.stride(0), n_out,
                                            _ptr(w), _ptr(b), cin, cout, 1 if relu else 0, _ptr(out),
                                            _ptr(head_w), _ptr(head_b), _ptr(logits)), "pcc_sparse_conv_head")
        return out, logits

    def sparse_conv_head_up(self, x, nbr_parent, w, b, relu, head_w, head_b):
        """sparse_conv_head on the 8 n generative children of a level, given THAT level's rule book [27, n]:
        the child rule book is formed inside the kernel (same bits as derive_map_up + sparse_conv_head)"""
        n_parents = nbr_parent.shape[1]
        assert x.shape == (8 * n_parents, 32) and w.shape == (27, 32, 32) and head_w.numel() == 32
        out = self.empty((8 * n_parents, 32), torch.float32)
        logits = self.empty((8 * n_parents,), torch.float32)
        check(self.lib.pcc_sparse_conv_head_up(self.ctx, _ptr(x), n_parents, _ptr(nbr_parent), nbr_parent.stride(0),
                                               _ptr(w), _ptr(b), 1 if relu else 0, _ptr(out), _ptr(head_w),
                                               _ptr(head_b), _ptr(logits)), "pcc_sparse_conv_head_up")
        return out, logits

    def conv_prepare(self, w):
        """register the weights [k, 32, 32|64] of a layer: re-arranged once, found by pointer afterwards (pcc_conv_prepare)"""
        check(self.lib.pcc_conv_prepare(self.ctx, _ptr(w), w.shape[0], w.shape[1], w.shape[2]), "pcc_conv_prepare")

    def conv_forget(self, w):
        check(self.lib.pcc_conv_forget(self.ctx, _ptr(w)), "pcc_conv_forget")

    def convT_gen(self, x, w, b, relu):
        n, cin, cout = x.shape[0], w.shape[1], w.shape[2]
        assert x.shape[1] == cin and w.shape[0] == 8
        out = self.empty((8 * n, cout), torch.float32)
        check(self.lib.pcc_convT_gen(self.ctx, _ptr(x), n, _ptr(w), _ptr(b), cin, cout, 1 if relu else 0,
                                     _ptr(out)), "pcc_convT_gen")
        return out

    def linear(self, x, w, b, relu):
        n, cin, cout = x.shape[0], w.shape[0], w.shape[1]
        assert x.shape[1] == cin
        out = self.empty((n, cout), torch.float32)
        check(self.lib.pcc_linear(self.ctx, _ptr(x), n, _ptr(w), _ptr(b), cin, cout, 1 if relu else 0, _ptr(out)),
              "pcc_linear")
        return out

    def convT_gen_gather(self, x, rows, w, b, relu):
        """convT_gen (32 -> 32) whose parent p is row rows[p] of x: [8 len(rows), 32]"""
        n = rows.shape[0]
        assert x.shape[1] == 32 and w.shape == (8, 32, 32)
        out = self.empty((8 * n, 32), torch.float32)
        check(self.lib.pcc_convT_gen_gather(self.ctx, _ptr(x), _ptr(rows), n, _ptr(w), _ptr(b), 1 if relu else 0,
                                            _ptr(out)), "pcc_convT_gen_gather")
        return out

    def linear_gather(self, x, rows, w, b, relu):
        """linear (cin 32, cout <= 8) on the rows `rows` (int32) of x: [len(rows), cout]"""
        n, cout = rows.shape[0], w.shape[1]
        assert x.shape[1] == 32 and w.shape[0] == 32
        out = self.empty((n, cout), torch.float32)
        check(self.lib.pcc_linear_gather(self.ctx, _ptr(x), _ptr(rows), n, _ptr(w), _ptr(b), cout, 1 if relu else 0,
                                         _ptr(out)), "pcc_linear_gather")
        return out

    def topk_prune(self, logits, offsets, k, with_map=False):
        """Kept rows (ascending); with_map: also remap [n] = position of a row among the kept ones, or -1."""
        n, nb = logits.shape[0], len(k)
        keep = self.empty((n,), torch.int32)
        offs = (C.c_int64 * (nb + 1))(*offsets)
        ks = (C.c_int64 * nb)(*k)
        nk = C.c_int64(0)
        if with_map:
            remap = self.empty((n,), torch.int32)
            check(self.lib.pcc_topk_prune_map(self.ctx, _ptr(logits), n, nb, offs, ks, _ptr(keep), C.byref(nk), _ptr(remap)),
                  "pcc_topk_prune_map")
            return keep[:nk.value], remap
        check(self.lib.pcc_topk_prune(self.ctx, _ptr(logits), n, nb, offs, ks, _ptr(keep), C.byref(nk)),
              "pcc_topk_prune")
        return keep[:nk.value]

    # ------------------------------------------------------------ entropy (device part)
    def factorized_quant(self, z, med):
        n, c = z.shape
        sym = self.empty((c, n), torch.int32)
        zhat = self.empty((n, c), torch.float32)
        check(self.lib.pcc_factorized_quant(self.ctx, _ptr(z), n, c, _ptr(med), _ptr(sym), _ptr(zhat)),
              "pcc_factorized_quant")
        return sym, zhat

    def factorized_dequant(self, sym, med):
        c, n = sym.shape
        zhat = self.empty((n, c), torch.float32)
        check(self.lib.pcc_factorized_dequant(self.ctx, _ptr(sym), n, c, _ptr(med), _ptr(zhat)),
              "pcc_factorized_dequant")
        return zhat

    def gaussian_quant(self, y, params, scale, table):
        n, c = y.shape
        q = scale.shape[0]
        sym = self.empty((q, c, n), torch.int32)
        idx = self.empty((q, c, n), torch.int32)
        check(self.lib.pcc_gaussian_quant(self.ctx, _ptr(y), _ptr(params), n, c, _ptr(scale), q, _ptr(table),
                                          table.shape[0], _ptr(sym), _ptr(idx)), "pcc_gaussian_quant")
        return sym, idx

    def build_indexes(self, scales, table):
        """element-wise GaussianConditional.build_indexes on a tensor of any shape"""
        scales = scales.contiguous()
        idx = self.empty(tuple(scales.shape), torch.int32)
        check(self.lib.pcc_build_indexes(self.ctx, _ptr(scales), scales.numel(), _ptr(table), table.shape[0],
                                         _ptr(idx)), "pcc_build_indexes")
        return idx

    def quantize_symbols(self, x, means=None):
        """element-wise round(x - means) -> int32 (EntropyModel.quantize(..., "symbols", means))"""
        x = x.contiguous()
        means = means.contiguous() if means is not None else None
        sym = self.empty(tuple(x.shape), torch.int32)
        check(self.lib.pcc_quantize_symbols(self.ctx, _ptr(x), _ptr(means), x.numel(), _ptr(sym)),
              "pcc_quantize_symbols")
        return sym

    def gaussian_quant16(self, y, params, scale, table):
        """compact form: int16 symbols, uint8 indexes, overflow flag (device int32[1])"""
        n, c = y.shape
        q = scale.shape[0]
        sym = self.empty((q, c, n), torch.int16)
        idx = self.empty((q, c, n), torch.uint8)
        flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        check(self.lib.pcc_gaussian_quant16(self.ctx, _ptr(y), _ptr(params), n, c, _ptr(scale), q, _ptr(table),
                                            table.shape[0], _ptr(sym), _ptr(idx), _ptr(flag)),
              "pcc_gaussian_quant16")
        return sym, idx, flag

    def gaussian_quant_dev(self, y, params, scale, table):
        """int32 symbols, uint8 indexes — the element types the GPU coder reads (pcc_gaussian_quant_dev)"""
        n, c = y.shape
        q = scale.shape[0]
        sym = self.empty((q, c, n), torch.int32)
        idx = self.empty((q, c, n), torch.uint8)
        check(self.lib.pcc_gaussian_quant_dev(self.ctx, _ptr(y), _ptr(params), n, c, _ptr(scale), q, _ptr(table),
                                              table.shape[0], _ptr(sym), _ptr(idx)), "pcc_gaussian_quant_dev")
        return sym, idx

    def gaussian_indexes8(self, params, scale, table):
        n, c = params.shape[0], params.shape[1] // 2
        idx = self.empty((c, n), torch.uint8)
        check(self.lib.pcc_gaussian_indexes8(self.ctx, _ptr(params), n, c, _ptr(scale), _ptr(table),
                                             table.shape[0], _ptr(idx)), "pcc_gaussian_indexes8")
        return idx

    # ------------------------------------------------------------ pinned staging
    def pinned(self, key, nbytes):
        """a cached page-locked host buffer of at least nbytes (uint8 tensor)"""
        pool = self.__dict__.setdefault("_pin", {})
        buf = pool.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 4096), dtype=torch.uint8, pin_memory=True)
            pool[key] = buf
        return buf

    def to_host_async(self, t, key):
        """device tensor -> numpy view of a pinned buffer; valid after the next sync()"""
        nbytes = t.numel() * t.element_size()
        buf = self.pinned(key, nbytes)[:nbytes].view(t.dtype).view(t.shape)
        buf.copy_(t, non_blocking=True)
        return buf.numpy()

    def from_pinned_async(self, host_tensor):
        """pinned host tensor -> new device tensor (async on this runtime's stream)"""
        return host_tensor.to(self.device, non_blocking=True)

    def gaussian_indexes(self, params, scale, table):
        n, c = params.shape[0], params.shape[1] // 2
        idx = self.empty((c, n), torch.int32)
        check(self.lib.pcc_gaussian_indexes(self.ctx, _ptr(params), n, c, _ptr(scale), _ptr(table),
                                            table.shape[0], _ptr(idx)), "pcc_gaussian_indexes")
        return idx

    def gaussian_dequant(self, sym, params, scale, bound, off_a, off_b):
        c, n = sym.shape
        yhat = self.empty((n, c), torch.float32)
        check(self.lib.pcc_gaussian_dequant(self.ctx, _ptr(sym), _ptr(params), n, c, _ptr(scale), bound, off_a,
                                            off_b, _ptr(yhat)), "pcc_gaussian_dequant")
        return yhat

    # ------------------------------------------------------------ octree (device part)
    def octree_levels(self, keys, key_shift, depth):
        n = keys.shape[0]
        cap = n * depth
        occ = self.empty((cap,), torch.uint8)
        level_n = (C.c_int64 * depth)()
        check(self.lib.pcc_octree_levels(self.ctx, _ptr(keys), n, key_shift, depth, _ptr(occ), cap, level_n),
              "pcc_octree_levels")
        ln = [int(v) for v in level_n]
        return occ[:sum(ln)], ln

    def octree_encode(self, keys, key_shift, version=0):
        """geometry blob of one frame's Morton-sorted keys (pcc_octree_encode_version): version 0 = the library's rule —
        blob version 2 (occupancy coder on the GPU, csrc/octree2.hip) above PCC_OCTREE_V2_MIN_LEAVES leaves, version 1
        (serial host coder) below"""
        n = keys.shape[0]
        cap = 4096 + 17 * max(n, 1)          # one byte per node at most, plus header and chunk table
        out = np.empty(cap, dtype=np.uint8)
        length = C.c_int64(0)
        check(self.lib.pcc_octree_encode_version(self.ctx, _ptr(keys), n, key_shift, version, _np_ptr(out), cap,
                                                 C.byref(length)), "pcc_octree_encode")
        return out[:length.value].tobytes()

    def octree_decode(self, blob):
        """blob (either version) -> int32 [n,3] host array, Morton order (pcc_octree_decode_ctx: version 2 is decoded
        by the GPU)"""
        buf = np.frombuffer(blob, dtype=np.uint8)
        n = C.c_int64(0)
        check(self.lib.pcc_octree_decode_ctx(self.ctx, _np_ptr(buf), buf.shape[0], None, 0, C.byref(n)), "pcc_octree_decode_ctx")
        pts = np.empty((n.value, 3), dtype=np.int32)
        if n.value:
            check(self.lib.pcc_octree_decode_ctx(self.ctx, _np_ptr(buf), buf.shape[0], _np_ptr(pts), n.value, C.byref(n)),
                  "pcc_octree_decode_ctx")
        return pts


OCTREE_V2_MIN_LEAVES = 65536     # include/pcc.h PCC_OCTREE_V2_MIN_LEAVES
OCTREE_V3_MIN_LEAVES = 8192      # include/pcc.h PCC_OCTREE_V3_MIN_LEAVES


# ---------------------------------------------------------------- GPU coder (container version 1)
class RansDev:
    """a CDF set in HBM for the interleaved rANS coder on the GPU (pcc_rans_dev_*, csrc/rans_gpu.hip)"""

    def __init__(self, cdfs, sizes, offsets):
        self.lib = _abi.lib()
        cdfs = np.ascontiguousarray(cdfs, dtype=np.int32)
        sizes = np.ascontiguousarray(sizes, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        self.handle = self.lib.pcc_rans_dev_create(_np_ptr(cdfs), cdfs.shape[1], _np_ptr(sizes), _np_ptr(offsets),
                                                   cdfs.shape[0])
        if not self.handle:
            raise PccError(-1, "pcc_rans_dev_create", self.lib.pcc_last_error().decode(errors="replace"))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pcc_rans_dev_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, rt, sym, idx=None, idx_run=1):
        """sym: int32 device tensor [S, n]; idx: uint8 device tensor [S, n] or None (table = position // idx_run).
        Returns the S streams as bytes."""
        assert sym.is_cuda and sym.dtype == torch.int32 and sym.dim() == 2 and sym.is_contiguous()
        s, n = sym.shape
        if idx is not None:
            assert idx.is_cuda and idx.dtype == torch.uint8 and idx.shape == sym.shape and idx.is_contiguous()
        cap = (int(self.lib.pcc_rans_dev_bound(n)) + 3) // 4 * 4
        cap = min(cap, 4 * (4 + n + 4096 + 130 * (n // 32768 + 1)) * 3)     # generous; the call reports if it is not
        out = rt.empty((s, cap), torch.uint8)
        lens = (C.c_int64 * s)()
        check(self.lib.pcc_rans_encode_dev(rt.ctx, self.handle, _ptr(sym), _ptr(idx) if idx is not None else None,
                                           idx_run, n, s, _ptr(out), cap, lens), "pcc_rans_encode_dev")
        host = out.cpu().numpy()
        return [host[i, :lens[i]].tobytes() for i in range(s)]

    def decode(self, rt, data, n, idx=None, idx_run=1):
        """data: bytes of one stream; returns the int32 device tensor [n] (raises on a malformed stream)"""
        buf = np.frombuffer(data, dtype=np.uint8)
        hn, ht, hc = C.c_int64(), C.c_int64(), C.c_int64()
        check(self.lib.pcc_rans_stream_info(_np_ptr(buf), buf.shape[0], C.byref(hn), C.byref(ht), C.byref(hc)),
              "pcc_rans_stream_info")
        if hn.value != n:
            raise PccError(-5, "RansDev.decode", f"stream holds {hn.value} symbols, {n} expected")
        d_in = rt.to_device(buf.copy())
        sym = rt.empty((n,), torch.int32)
        status = torch.zeros(1, dtype=torch.int32, device=sym.device)
        check(self.lib.pcc_rans_decode_dev(rt.ctx, self.handle, _ptr(d_in), buf.shape[0], n, ht.value, hc.value,
                                           _ptr(idx) if idx is not None else None, idx_run, _ptr(sym), _ptr(status)),
              "pcc_rans_decode_dev")
        rt.sync()
        st = int(status.item())
        if st:
            raise PccError(-5, "pcc_rans_decode_dev", f"malformed interleaved rANS stream (status {st})")
        return sym


# ---------------------------------------------------------------- host coders (no ctx)
def rans_encode_multi(sym, idx, cdfs, sizes, offsets):
    """sym/idx: numpy [S, n], either (int32, int32) or the compact (int16, uint8);
    returns list of S byte strings"""
    lib = _abi.lib()
    s, n = sym.shape
    if sym.dtype == np.int16 and idx.dtype == np.uint8:
        fn, name = lib.pcc_rans_encode_multi16, "pcc_rans_encode_multi16"
    elif sym.dtype == np.int32 and idx.dtype == np.int32:
        fn, name = lib.pcc_rans_encode_multi, "pcc_rans_encode_multi"
    else:
        raise TypeError(f"rans_encode_multi: unsupported dtypes {sym.dtype}/{idx.dtype}")
    lens = (C.c_int64 * s)()
    cap = 2 * n + 4096                      # in-range symbols cost <= 16 bits each
    for _ in range(2):
        out = np.empty((s, cap), dtype=np.uint8)
        rc = fn(_np_ptr(sym), _np_ptr(idx), n, s, _np_ptr(cdfs), cdfs.shape[1], _np_ptr(sizes), _np_ptr(offsets),
                cdfs.shape[0], _np_ptr(out), cap, lens)
        if rc != _abi.PCC_E_NOMEM:
            break
        cap = 48 * n + 4096                 # escape-heavy input: absolute worst case
    check(rc, name)
    return [out[i, :lens[i]].tobytes() for i in range(s)]


def rans_encode(sym, idx, cdfs, sizes, offsets):
    return rans_encode_multi(sym.reshape(1, -1), idx.reshape(1, -1), cdfs, sizes, offsets)[0]


def rans_decode(data, idx, cdfs, sizes, offsets, out=None):
    """idx: int32 or uint8 numpy [n]; returns int32 symbols (written into `out` if given)"""
    lib = _abi.lib()
    n = idx.shape[0]
    buf = np.frombuffer(data, dtype=np.uint8)
    sym = out if out is not None else np.empty(n, dtype=np.int32)
    assert sym.dtype == np.int32 and sym.shape[0] == n
    if idx.dtype == np.uint8:
        fn, name = lib.pcc_rans_decode8, "pcc_rans_decode8"
    else:
        fn, name = lib.pcc_rans_decode, "pcc_rans_decode"
        assert idx.dtype == np.int32
    check(fn(_np_ptr(buf), buf.shape[0], _np_ptr(idx), n, _np_ptr(cdfs), cdfs.shape[1], _np_ptr(sizes),
             _np_ptr(offsets), cdfs.shape[0], _np_ptr(sym)), name)
    return sym


def octree_pack(occ, level_n, n_points, origin):
    lib = _abi.lib()
    depth = len(level_n)
    cap = 64 + 2 * len(occ) + 16
    out = np.empty(cap, dtype=np.uint8)
    ln = (C.c_int64 * max(depth, 1))(*level_n)
    org = (C.c_int * 3)(*[int(v) for v in origin])
    length = C.c_int64(0)
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    check(lib.pcc_octree_pack(_np_ptr(occ) if len(occ) else C.c_void_p(0), ln, depth, n_points, org,
                              _np_ptr(out), cap, C.byref(length)), "pcc_octree_pack")
    return out[:length.value].tobytes()


def octree_unpack(blob):
    lib = _abi.lib()
    buf = np.frombuffer(blob, dtype=np.uint8)
    n = C.c_int64(0)
    depth = C.c_int(0)
    org = (C.c_int * 3)()
    check(lib.pcc_octree_peek(_np_ptr(buf), buf.shape[0], C.byref(n), C.byref(depth), org), "pcc_octree_peek")
    pts = np.empty((n.value, 3), dtype=np.int32)
    if n.value:
        check(lib.pcc_octree_unpack(_np_ptr(buf), buf.shape[0], _np_ptr(pts), n.value), "pcc_octree_unpack")
    return pts
