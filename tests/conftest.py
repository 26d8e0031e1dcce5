import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "demo-learned-point-cloud-compression_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    return importlib.import_module(PKG + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def oracle():
    from oracle.codec_ref import Oracle
    # size the oracle's OpenMP pool to the cgroup share (the GPU boxes show 256 cores, grant 16)
    return Oracle(threads=min(8, pkg("_abi").host_cpu_budget()))


@pytest.fixture(scope="session")
def wl():
    return pkg("workloads")


@pytest.fixture(scope="session")
def rt():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    runtime = pkg("runtime")
    r = runtime.Runtime(0)
    r.__enter__()
    yield r
    r.__exit__(None, None, None)
    r.close()


def random_cloud(rng, n, extent=64, batches=1, lo=-40, stride=1):
    """unique voxel coordinates [n,4] (b,x,y,z), multiples of `stride`, some negative"""
    pts = set()
    out = []
    while len(out) < n:
        b = int(rng.integers(0, batches))
        p = tuple(int(v) for v in rng.integers(0, extent, 3))
        if (b,) + p in pts:
            continue
        pts.add((b,) + p)
        out.append((b, (p[0] + lo) * stride, (p[1] + lo) * stride, (p[2] + lo) * stride))
    return np.asarray(out, dtype=np.int32)


def surface_cloud(rng, n, batches=1, stride=1):
    """points on a few random planes/spheres: surface-like neighbourhood statistics"""
    out = []
    per = n // batches
    for b in range(batches):
        pts = []
        for _ in range(3):
            c = rng.integers(-20, 20, 3)
            r = rng.integers(8, 20)
            v = rng.normal(size=(per, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            pts.append(np.rint(v * r + c).astype(np.int64))
        p = np.unique(np.concatenate(pts, 0), axis=0)[:per]
        out.append(np.concatenate([np.full((p.shape[0], 1), b), p * stride], 1))
    return np.concatenate(out, 0).astype(np.int32)
