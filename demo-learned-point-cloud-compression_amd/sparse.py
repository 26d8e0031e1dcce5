"""SparseTensor / CoordSet — the slice of MinkowskiEngine's tensor surface that
the reference codec uses: ctor kwargs `coordinates, features,
tensor_stride, device`; attributes `.C` (int32 [N,4] = b,x,y,z), `.F`
(float32 [N,C]), `.tensor_stride`, `.device`; `.features_at_coordinates()`.

Rows are stored sorted by Morton key (include/pcc.h).  A CoordSet is the
analogue of a coordinate-map key in ME's coordinate manager: tensors that share
coordinates share one CoordSet and therefore its cached rule books, parents and
children.

Rule books (3^3 kernel maps) are built with the coordinate hash only at the
coarsest level of a pyramid; every finer level derives its rule book from its
parent's with table lookups (csrc/map.hip, pcc_derive_map_*):
  - generative children (`up()`), also when the parent level is a top-k pruned
    subset of its own candidate level (`subset()`);
  - levels linked by a stride-2 map (`down()`).
"""
import weakref

import numpy as np
import torch

from . import runtime as _rt

HASH_BUILD_MAX = 60000      # levels up to this many voxels hash directly; larger ones derive


def _log2(ts):
    s = int(ts).bit_length() - 1
    if ts < 1 or (1 << s) != ts:
        raise ValueError(f"tensor_stride must be a power of two, got {ts}")
    return s


class CoordSet:
    def __init__(self, rt, keys, stride, n_batch=None, offsets=None):
        self.rt = rt
        self.keys = keys                 # int64 tensor holding uint64 Morton keys, sorted, unique
        self.stride = int(stride)
        self.n = int(keys.shape[0])
        self._coords = None
        self._nbr27 = None
        self._down = None                # (CoordSet, nbr8, parent_of)
        self._up = None                  # CoordSet
        self._gen_parent = None          # CoordSet whose generative children these rows are
        self._subset_of = None           # (candidate CoordSet, keep rows) for a pruned level
        self._n_batch = n_batch
        self._offsets = offsets          # host list, len n_batch+1

    @property
    def C(self):
        if self._coords is None:
            self._coords = self.rt.keys_to_coords(self.keys)
        return self._coords

    def set_batches(self, n_batch, offsets=None):
        self._n_batch = int(n_batch)
        self._offsets = offsets

    @property
    def n_batch(self):
        if self._n_batch is None:
            # batch index lives in the top 16 key bits; the last row has the largest
            self._n_batch = (int(self.keys[-1].item()) >> 48) + 1 if self.n else 0
        return self._n_batch

    @property
    def offsets(self):
        """row offsets per batch index (host list of n_batch+1 ints)"""
        if self._offsets is None:
            if self.n_batch == 1:
                self._offsets = [0, self.n]      # one frame: nothing to look up on the device
            else:
                self._offsets = self.rt.batch_offsets(self.keys, self.n_batch)
        return self._offsets

    # ------------------------------------------------------------ rule book
    def nbr27(self):
        if self._nbr27 is None:
            self._nbr27 = self._build_nbr27()
            log = getattr(self.rt, "pairs_log", None)
            if log is not None:     # bench.py: active-pair count of each rule book, keyed by row count
                log[self.n] = self.rt.count_nonneg(self._nbr27)
        return self._nbr27

    def _build_nbr27(self):
        rt = self.rt
        gp = self._gen_parent
        if gp is not None and gp.n > 0:
            if gp._nbr27 is None and gp._subset_of is not None:
                # parent is a pruned subset of its candidate level: go through the candidates' rule book
                cand, keep = gp._subset_of
                remap = rt.inverse_rows(keep, cand.n)
                return rt.derive_map_up(cand.nbr27(), gp.n, keep, remap)
            return rt.derive_map_up(gp.nbr27(), gp.n)
        if self.n > HASH_BUILD_MAX and self.stride <= 4096:
            pcs, nbr8, parent_of = self.down()
            return rt.derive_map_down(pcs.nbr27(), nbr8, parent_of, self.keys, 3 * _log2(self.stride))
        return rt.build_map(self.keys, self.stride)

    # ------------------------------------------------------------ pyramid
    def down(self):
        """parents at stride*2, the kernel-2 rule book [8, M] and the parent row of every row"""
        if self._down is None:
            pkeys, nbr8, parent_of = self.rt.down_coords(self.keys, 3 * _log2(self.stride))
            self._down = (CoordSet(self.rt, pkeys, self.stride * 2, self._n_batch), nbr8, parent_of)
        return self._down

    def up(self):
        """generative children at stride/2: 8 per row, row 8p+o"""
        child = self._up() if self._up is not None else None
        if child is None:
            if self.stride < 2:
                raise ValueError("cannot up-sample a stride-1 coordinate set")
            ckeys = self.rt.up_coords(self.keys, 3 * (_log2(self.stride) - 1))
            offs = [8 * o for o in self._offsets] if self._offsets is not None else None
            child = CoordSet(self.rt, ckeys, self.stride // 2, self._n_batch, offs)
            child._gen_parent = self
            # weak: the child keeps its parent alive (it derives its rule book from it), not the
            # other way round — a strong cycle would leave GBs of HBM to the cyclic GC
            self._up = weakref.ref(child)
        return child

    def subset(self, rows, n_batch=None, offsets=None):
        """stable compaction: keep `rows` (ascending uint32 indices)"""
        cs = CoordSet(self.rt, self.rt.gather_rows(self.keys, rows), self.stride,
                      n_batch if n_batch is not None else self._n_batch, offsets)
        cs._subset_of = (self, rows)
        return cs


class SparseTensor:
    """ME.SparseTensor-shaped.  `coordinates` may be a float or int tensor /
    array [N,4] (b,x,y,z); floats are floored like ME does (the reference
    passes float coordinates, codec_pipeline.py:259-266)."""

    def __init__(self, features=None, coordinates=None, tensor_stride=1, device=None, coordset=None, rt=None):
        rt = rt if rt is not None else (coordset.rt if coordset is not None else _rt.current())
        self.rt = rt
        if coordset is not None:
            self.cs = coordset
            self.F = features
            return
        ts = tensor_stride[0] if isinstance(tensor_stride, (list, tuple)) else int(tensor_stride)
        _log2(ts)
        coords = coordinates
        if isinstance(coords, np.ndarray):
            coords = torch.from_numpy(np.ascontiguousarray(coords))
        if coords.dtype.is_floating_point:
            coords = torch.floor(coords)
        coords = rt.to_device(coords, torch.int32)
        feats = rt.to_device(features, torch.float32)
        if coords.ndim != 2 or coords.shape[1] != 4 or feats.shape[0] != coords.shape[0]:
            raise ValueError(f"coordinates must be [N,4] and match features, got {tuple(coords.shape)} / "
                             f"{tuple(feats.shape)}")
        keys = rt.morton_keys(coords)            # raises PccError(RANGE) on out-of-range input
        perm = rt.sort_pairs(keys)               # keys sorted in place
        if not rt.check_unique(keys):
            raise _rt.PccError(-4, "SparseTensor", "duplicate coordinates")
        self.cs = CoordSet(rt, keys, ts)
        self.F = rt.gather_rows(feats, perm)

    # --- ME surface -----------------------------------------------------
    @property
    def C(self):
        return self.cs.C

    @property
    def tensor_stride(self):
        return [self.cs.stride] * 3

    @property
    def device(self):
        return self.rt.device

    @property
    def shape(self):
        return self.F.shape

    def features_at_coordinates(self, query):
        """exact-lattice lookup, zeros where the coordinate is absent
        (codec_pipeline.py:401, codec_parallel.py:387).  query: [M,4] float/int."""
        q = query
        if q.dtype.is_floating_point:
            q = torch.floor(q)
        q = self.rt.to_device(q, torch.int32)
        qkeys = self.rt.morton_keys(q)
        rows = self.rt.lookup(self.cs.keys, qkeys)
        return self.rt.gather_rows_or_zero(self.F, rows)
