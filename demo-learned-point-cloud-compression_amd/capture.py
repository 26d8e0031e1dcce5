"""GPU voxelisation of a camera frame — the step immediately before the codec path
(reference: sender/capturer/capturer.py:88-126, numpy + Open3D on the host).

    voxelize(rt, xyzrgba, depth_clip, voxel_size, max_points) -> {"points": int16 [N,3], "colors": float64 [N,3]}

xyzrgba: float32 [M,4] as delivered by the ZED SDK (`sl.MEASURE.XYZRGBA`): metres, colour packed in the
bits of the 4th float (r = bits 0-7, g = 8-15, b = 16-23).  Steps, all on the device through the C-ABI:
valid mask + per-axis minimum (`pcc_vox_valid`), Open3D voxel index (`pcc_vox_keys`), stable radix sort by
voxel (`pcc_sort_pairs`), per-voxel mean in double + integer voxel (`pcc_vox_mean`), canonical sort and
adjacent-unique on the integer voxels (`pcc_sort_coords`, `pcc_unique_rows`), optional cap to the
`max_points` largest z (`pcc_topk_prune`).

Where the reference's result depends on an unspecified order this build fixes one:
duplicates after rounding keep the voxel whose Open3D index is smallest (x-major); at the max_points
threshold ties in z go to the row that comes first in (x,y,z) order; output rows are in (x,y,z) order
(what np.unique returns in the reference).
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._abi import check
from .runtime import _ptr


def voxelize(rt, xyzrgba, depth_clip=1.4, voxel_size=0.005, max_points=None, output="numpy"):
    lib = rt.lib
    data = rt.to_device(xyzrgba, torch.float32)
    if data.dim() != 2 or data.shape[1] != 4:
        raise ValueError(f"xyzrgba must be [M,4] float32, got {tuple(data.shape)}")
    m = int(data.shape[0])
    empty = {"points": np.zeros((0, 3), np.int16), "colors": np.zeros((0, 3), np.float64)}
    if m == 0:
        return empty
    # 1. valid mask, min bound
    valid = rt.empty((m,), torch.uint8)
    mn = (C.c_float * 3)()
    n_valid = C.c_int64(0)
    check(lib.pcc_vox_valid(rt.ctx, _ptr(data), m, float(np.float32(depth_clip)), _ptr(valid), mn,
                            C.byref(n_valid)), "pcc_vox_valid")
    n_valid = n_valid.value
    if n_valid == 0:
        return empty
    vs = float(voxel_size)
    # Open3D: voxel_min_bound = min_bound - voxel_size * 0.5 (double)
    vmb = (C.c_double * 3)(*[float(np.float64(mn[a]) - np.float64(vs) * 0.5) for a in range(3)])
    # 2. voxel keys, 3. stable sort (invalid points carry the all-ones key and end up last)
    keys = rt.empty((m,), torch.int64)
    flag = torch.zeros(1, dtype=torch.int32, device=rt.device)
    check(lib.pcc_vox_keys(rt.ctx, _ptr(data), _ptr(valid), m, vmb, vs, _ptr(keys), _ptr(flag)), "pcc_vox_keys")
    perm = rt.sort_pairs(keys)
    if int(flag.item()) != 0:
        raise _abi.PccError(_abi.PCC_E_RANGE, "voxelize", "scene extent / voxel_size exceeds 2^21 voxels per axis")
    # 4. per-voxel mean -> integer voxel rows (0,x,y,z), float64 colours
    coords = rt.empty((n_valid, 4), torch.int32)
    colors = rt.empty((n_valid, 3), torch.float64)
    n_vox = C.c_int64(0)
    check(lib.pcc_vox_mean(rt.ctx, _ptr(data), _ptr(keys), _ptr(perm), n_valid, vs, _ptr(coords), _ptr(colors),
                           n_valid, C.byref(n_vox)), "pcc_vox_mean")
    n_vox = n_vox.value
    coords, colors = coords[:n_vox], colors[:n_vox]
    # 5. duplicates after rounding: canonical (x,y,z) sort is stable, so the first row of a run is the
    #    voxel with the smallest Open3D index
    p2 = rt.sort_coords(coords)
    coords = rt.gather_rows(coords, p2)
    colors = rt.gather_rows(colors, p2)
    rows = rt.empty((n_vox,), torch.int32)
    n_u = C.c_int64(0)
    check(lib.pcc_unique_rows(rt.ctx, _ptr(coords), n_vox, _ptr(rows), C.byref(n_u)), "pcc_unique_rows")
    rows = rows[:n_u.value]
    coords = rt.gather_rows(coords, rows)
    colors = rt.gather_rows(colors, rows)
    # 6. cap: the max_points largest z (capturer.py:119-122)
    if max_points is not None and coords.shape[0] > max_points:
        z = coords[:, 3].to(torch.float32).contiguous()          # |z| < 2^24: exact
        keep = rt.topk_prune(z, [0, int(coords.shape[0])], [int(max_points)])
        coords = rt.gather_rows(coords, keep)
        colors = rt.gather_rows(colors, keep)
    if output == "device":
        return {"points": coords[:, 1:], "colors": colors}
    pts = coords[:, 1:].cpu().numpy()
    if pts.size and (pts.min() < -32768 or pts.max() > 32767):
        raise _abi.PccError(_abi.PCC_E_RANGE, "voxelize", "integer voxel outside int16 (capturer.py:107)")
    return {"points": pts.astype(np.int16), "colors": colors.cpu().numpy()}
