"""Op-level parity: libpcc_hip.so (through the C-ABI) vs the CPU oracle on the
same seeded inputs.  Integer / index work must match bit for bit; float layers
must match bit for bit too (both sides are the fmaf chain of include/pcc.h)."""
import os

import numpy as np
import pytest

from conftest import pkg, random_cloud, surface_cloud

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(rt, a):
    return rt.to_device(np.ascontiguousarray(a))


def host(t):
    return t.cpu().numpy()


def u64(t):
    return host(t).view(np.uint64)


@pytest.fixture(scope="module")
def clouds():
    rng = np.random.default_rng(7)
    return {
        "rand": random_cloud(rng, 5000, extent=40, batches=3),
        "surf": surface_cloud(rng, 6000, batches=2),
        "tiny": random_cloud(rng, 37, extent=6, batches=1),
        "one": np.array([[0, -5, 7, 9]], dtype=np.int32),
    }


def sorted_keys(oracle, coords):
    return np.sort(oracle.morton_keys(coords))


# ---------------------------------------------------------------- keys / sort
@pytest.mark.parametrize("name", ["rand", "surf", "tiny", "one"])
def test_morton_keys_and_inverse(rt, oracle, clouds, name):
    c = clouds[name]
    k = rt.morton_keys(dev(rt, c))
    assert np.array_equal(u64(k), oracle.morton_keys(c))
    back = rt.keys_to_coords(k)
    assert np.array_equal(host(back), c)


def test_morton_range_error(rt):
    runtime = pkg("runtime")
    bad = np.array([[0, 40000, 0, 0]], dtype=np.int32)
    with pytest.raises(runtime.PccError) as e:
        rt.morton_keys(dev(rt, bad))
    assert e.value.code == -3


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 1024, 1025, 40000])
def test_sort_pairs_matches_stable_argsort(rt, n):
    rng = np.random.default_rng(n)
    # few distinct values in some digits -> exercises pass skipping and stability
    keys = (rng.integers(0, 1 << 20, n).astype(np.uint64) << np.uint64(17)) | rng.integers(0, 4, n).astype(np.uint64)
    k = dev(rt, keys.view(np.int64))
    perm = rt.sort_pairs(k)
    ref = np.argsort(keys, kind="stable")
    assert np.array_equal(host(perm).view(np.uint32), ref.astype(np.uint32))
    assert np.array_equal(u64(k), keys[ref])


@pytest.mark.parametrize("case", ["few_high_parts", "many_high_parts", "one_high_part", "signed_few", "straddling_zero"])
def test_sort_pairs_multi_kernel_path(rt, oracle, case):
    """n above the single-workgroup limit: the multi-kernel radix passes (constant bytes skipped), stable in every case
    (repeated keys keep their input order)"""
    rng = np.random.default_rng(len(case))
    n = 150_001
    signed = case == "signed_few"
    low = rng.integers(0, 1 << 24, n).astype(np.uint64)
    low[::7] = low[0]                                   # repeated keys
    if case == "few_high_parts":
        hi = rng.choice(rng.integers(0, 1 << 38, 200), n).astype(np.uint64)
    elif case == "many_high_parts":
        hi = rng.integers(0, 1 << 38, n).astype(np.uint64)
    elif case == "one_high_part":
        hi = np.full(n, 0x12345, np.uint64)
    elif case == "signed_few":
        hi = rng.choice(np.array([0, 1, 5, (1 << 40) - 1, (1 << 40) - 9, 1 << 39], np.uint64), n)
    else:
        pts = rng.integers(-300, 300, (n, 3))
        c = np.concatenate([rng.integers(0, 3, (n, 1)), pts], 1).astype(np.int32)
        keys = host(rt.morton_keys(dev(rt, c))).view(np.uint64)
        hi = low = None
    if hi is not None:
        keys = (hi << np.uint64(24)) | low
    ref = np.argsort(keys.view(np.int64) if signed else keys, kind="stable")
    k = dev(rt, keys.view(np.int64))
    perm = rt.sort_pairs(k, signed=signed)
    assert np.array_equal(host(perm).view(np.uint32), ref.astype(np.uint32))
    assert np.array_equal(u64(k), keys[ref])


@pytest.mark.parametrize("n", [65537, 66000, 100_003, 131072, 131073])
@pytest.mark.parametrize("signed", [False, True])
def test_sort_pairs_just_above_the_single_workgroup_limit(rt, n, signed):
    """sorts of 65 537 .. 131 073 keys: the multi-kernel radix passes on few wave tiles — a last workgroup with one live
    wave (66 000 keys = 65 tiles), 128 and 129 tiles; stable, signed and unsigned.  (A form of these passes in which the
    scatter kernel scans the raw digit counts itself — two launches per pass instead of four — passed this test and was
    slower: 185 against 142 us for 95k keys)"""
    rng = np.random.default_rng(n + int(signed))
    keys = rng.integers(-(1 << 46), 1 << 46, n, dtype=np.int64) if signed else rng.integers(0, 1 << 47, n, dtype=np.int64)
    keys[::5] = keys[3]                                 # repeated keys: stability
    keys[1::2] &= ~np.int64(0xFF00)                     # a digit with few distinct values
    k = dev(rt, keys.copy())
    perm = rt.sort_pairs(k, signed=signed)
    ref = np.argsort(keys if signed else keys.view(np.uint64), kind="stable")
    assert np.array_equal(host(perm).view(np.uint32), ref.astype(np.uint32))
    assert np.array_equal(host(k), keys[ref])


def test_sort_pairs_signed(rt):
    rng = np.random.default_rng(3)
    keys = rng.integers(-(1 << 62), 1 << 62, 5000, dtype=np.int64)
    k = dev(rt, keys)
    perm = rt.sort_pairs(k, signed=True)
    ref = np.argsort(keys, kind="stable")
    assert np.array_equal(host(perm).view(np.uint32), ref.astype(np.uint32))


@pytest.mark.parametrize("name", ["rand", "surf", "tiny"])
def test_sort_coords_is_reference_order(rt, oracle, clouds, name):
    """shared/utils.py:131-133: argsort of b*1e15 + x*1e10 + y*1e5 + z"""
    c = clouds[name]
    perm = host(rt.sort_coords(dev(rt, c))).view(np.uint32)
    assert np.array_equal(perm, oracle.canonical_perm(c).astype(np.uint32))
    # and it is the lexicographic order for int16-range coordinates
    srt = c[perm]
    assert all(tuple(srt[i]) < tuple(srt[i + 1]) for i in range(len(srt) - 1))


@pytest.mark.parametrize("case", ["latent", "one", "two", "dups", "wide", "wide_dups", "out_of_range", "max_rows"])
def test_sort_coords_routes(rt, oracle, case):
    """the three routes of the latent-sized canonical sort: rank by presence bitmap (distinct rows, key domain
    <= 2^20), radix passes on the compact key (repeated rows or a wider domain; the order among equal rows must be
    the stable one), radix passes on the decimal key (coordinates outside +-50000)"""
    rng = np.random.default_rng(11)
    if case == "latent":      # stride-8 latent of a few frames, some negative: ~18-bit domain
        c = random_cloud(rng, 20000, extent=48, batches=3, lo=-20, stride=8)
    elif case == "one":
        c = np.array([[2, -8, 16, 24]], np.int32)
    elif case == "two":
        c = np.array([[0, 8, 0, 0], [0, -8, 0, 0]], np.int32)
    elif case == "dups":
        c = random_cloud(rng, 3000, extent=20, batches=2, lo=-5, stride=8)
        c = np.concatenate([c, c[::7], c[::13]])[rng.permutation(3000 + len(c[::7]) + len(c[::13]))]
    elif case == "wide":      # distinct rows, domain of ~2^27 keys
        c = random_cloud(rng, 5000, extent=512, batches=1, lo=-256, stride=1)
    elif case == "wide_dups":
        c = random_cloud(rng, 2000, extent=512, batches=2, lo=-256, stride=1)
        c = np.concatenate([c, c[:500]])
    elif case == "out_of_range":
        c = random_cloud(rng, 2000, extent=64, batches=2, lo=-32, stride=1)
        c[:, 1] *= 2000
    else:                     # 65536 rows: the largest single-workgroup case, 2^16 of a 2^18-key domain
        g = np.stack(np.meshgrid(np.arange(64), np.arange(64), np.arange(64), indexing="ij"), -1).reshape(-1, 3)
        g = g[rng.permutation(len(g))[:65536]] * 4 - 100
        c = np.concatenate([np.zeros((len(g), 1), np.int64), g], 1).astype(np.int32)
    perm = host(rt.sort_coords(dev(rt, c))).view(np.uint32)
    assert np.array_equal(perm, oracle.canonical_perm(c).astype(np.uint32))


def test_batch_offsets(rt, oracle, clouds):
    keys = sorted_keys(oracle, clouds["rand"])
    offs = rt.batch_offsets(dev(rt, keys.view(np.int64)), 3)
    assert offs == oracle.batch_offsets(keys, 3)


@pytest.mark.parametrize("n", [1, 7, 2047, 2048, 2049, 4096, 100_003, 3_262_640, 20_000_000])
def test_exclusive_scan(rt, n):
    """sizes from one tile to ~10k tiles (one to three levels of recursion), repeated on the same ctx, in place
    and out of place, with values that wrap mod 2^32"""
    rng = np.random.default_rng(n)
    for rep in range(3):
        hi = 3 if rep < 2 else 1 << 31
        a = rng.integers(0, hi, n, dtype=np.uint32)
        ref = np.concatenate([np.zeros(1, np.uint64), np.cumsum(a.astype(np.uint64))]) & np.uint64(0xFFFFFFFF)
        d = dev(rt, a.view(np.int32))
        out, tot = rt.exclusive_scan(d, inplace=(rep == 1))
        assert np.array_equal(host(out).view(np.uint32), ref[:-1].astype(np.uint32))
        assert host(tot).view(np.uint32)[0] == np.uint32(ref[-1])


# ---------------------------------------------------------------- coordinate maps
@pytest.mark.parametrize("name,stride", [("rand", 1), ("surf", 1), ("tiny", 1), ("one", 1), ("surf", 4)])
def test_down_coords(rt, oracle, clouds, name, stride):
    c = clouds[name].copy()
    c[:, 1:] *= stride
    keys = sorted_keys(oracle, c)
    pk, nbr8, parent_of = rt.down_coords(dev(rt, keys.view(np.int64)), 3 * (stride.bit_length() - 1))
    rpk, rnbr = oracle.down(keys, stride)
    assert np.array_equal(u64(pk), rpk)
    assert np.array_equal(host(nbr8), rnbr)
    shift = np.uint64(3 * stride.bit_length())
    assert np.array_equal(rpk[host(parent_of)], (keys >> shift) << shift)


@pytest.mark.parametrize("name,stride", [("rand", 1), ("surf", 1), ("tiny", 1), ("one", 1), ("surf", 8)])
def test_level_counts_are_the_pyramid_sizes(rt, oracle, clouds, name, stride):
    """pcc_level_counts: the sizes of five successive stride-2 parent sets (batch index included in the key) from one
    pass, and pcc_down_coords_known fed with them == pcc_down_coords; repeated rows raise the duplicate flag"""
    c = clouds[name].copy()
    c[:, 1:] *= stride
    keys = sorted_keys(oracle, c)
    kd = dev(rt, keys.view(np.int64))
    cshift = 3 * (stride.bit_length() - 1)
    counts, dup = rt.level_counts(kd, cshift, 5)
    assert not dup
    cur, s = keys, stride
    for lvl in range(5):
        pk, nbr8 = oracle.down(cur, s)
        assert counts[lvl] == len(pk)
        if lvl < 2:
            d = dev(rt, cur.view(np.int64))
            a = rt.down_coords(d, 3 * (s.bit_length() - 1))
            b = rt.down_coords(d, 3 * (s.bit_length() - 1), m_known=counts[lvl])
            for x, y in zip(a, b):
                assert np.array_equal(host(x), host(y))
        cur, s = pk, s * 2
    if len(keys) > 1:
        twice = np.sort(np.concatenate([keys, keys[:1]]))
        assert rt.level_counts(dev(rt, twice.view(np.int64)), cshift, 1)[1]


@pytest.mark.parametrize("name,stride", [("surf", 1), ("rand", 2), ("tiny", 1)])
def test_derived_map_down_equals_hash_map(rt, oracle, clouds, name, stride):
    """rule book derived from the parent level == rule book from the coordinate hash == oracle"""
    c = clouds[name].copy()
    c[:, 1:] *= stride
    keys = sorted_keys(oracle, c)
    kd = dev(rt, keys.view(np.int64))
    cshift = 3 * (stride.bit_length() - 1)
    pk, nbr8, parent_of = rt.down_coords(kd, cshift)
    nbr_p = rt.build_map(pk, stride * 2)
    nbr = rt.derive_map_down(nbr_p, nbr8.contiguous(), parent_of, kd, cshift)
    assert np.array_equal(host(nbr), oracle.map27(keys, stride))


@pytest.mark.parametrize("prune", [False, True])
def test_derived_map_up_equals_oracle(rt, oracle, clouds, prune):
    """generative children of a (possibly pruned) parent level"""
    c = clouds["surf"].copy()
    c[:, 1:] *= 4
    cand_keys = sorted_keys(oracle, c)                       # plays the candidate level (stride 4)
    cand = dev(rt, cand_keys.view(np.int64))
    nbr_cand = rt.build_map(cand, 4)
    if prune:
        rng = np.random.default_rng(0)
        keep = np.sort(rng.permutation(len(cand_keys))[: len(cand_keys) // 2]).astype(np.uint32)
        keep_d = dev(rt, keep.view(np.int32))
        remap = rt.inverse_rows(keep_d, len(cand_keys))
        par_keys = cand_keys[keep]
        nbr = rt.derive_map_up(nbr_cand, len(keep), keep_d, remap)
    else:
        par_keys = cand_keys
        nbr = rt.derive_map_up(nbr_cand, len(cand_keys))
    child_keys = oracle.up(par_keys, 4)
    assert np.array_equal(host(nbr), oracle.map27(child_keys, 2))


def test_up_coords(rt, oracle, clouds):
    c = clouds["tiny"].copy()
    c[:, 1:] *= 8
    keys = sorted_keys(oracle, c)
    ck = rt.up_coords(dev(rt, keys.view(np.int64)), 6)
    assert np.array_equal(u64(ck), oracle.up(keys, 8))
    assert np.all(np.diff(u64(ck).astype(np.float64)) > 0)  # children come out Morton-sorted
    # the keys of listed children alone (what a pruning keeps), without the 8 n keys being formed
    rng = np.random.default_rng(5)
    rows = np.sort(rng.choice(8 * len(keys), size=3 * len(keys), replace=False)).astype(np.uint32)
    rows[:8] = np.arange(8 * len(keys) - 8, 8 * len(keys))      # the last parent's eight, out of order with the rest
    sub = rt.up_coords_rows(dev(rt, keys.view(np.int64)), 6, dev(rt, rows.view(np.int32)))
    assert np.array_equal(u64(sub), oracle.up(keys, 8)[rows.astype(np.int64)])
    assert rt.up_coords_rows(dev(rt, keys.view(np.int64)), 6, dev(rt, rows[:0].view(np.int32))).shape[0] == 0


@pytest.mark.parametrize("name,stride", [("rand", 1), ("surf", 1), ("tiny", 1), ("one", 1), ("surf", 8)])
def test_build_map(rt, oracle, clouds, name, stride):
    c = clouds[name].copy()
    c[:, 1:] *= stride
    keys = sorted_keys(oracle, c)
    nbr = rt.build_map(dev(rt, keys.view(np.int64)), stride)
    assert np.array_equal(host(nbr), oracle.map27(keys, stride))


def test_build_map_edge_of_range(rt, oracle):
    c = np.array([[0, 32767, 32767, 32767], [0, 32766, 32767, 32767], [0, -32768, -32768, -32768],
                  [1, -32768, -32768, -32767]], dtype=np.int32)
    keys = sorted_keys(oracle, c)
    nbr = rt.build_map(dev(rt, keys.view(np.int64)), 1)
    assert np.array_equal(host(nbr), oracle.map27(keys, 1))


def test_lookup(rt, oracle, clouds):
    keys = sorted_keys(oracle, clouds["surf"])
    rng = np.random.default_rng(5)
    q = np.concatenate([keys[rng.integers(0, len(keys), 500)], oracle.morton_keys(random_cloud(rng, 300, 30, 2))])
    rows = rt.lookup(dev(rt, keys.view(np.int64)), dev(rt, q.view(np.int64)))
    assert np.array_equal(host(rows), oracle.lookup(keys, q))
    feats = rng.normal(size=(len(keys), 8)).astype(np.float32)
    g = rt.gather_rows_or_zero(dev(rt, feats), rows)
    ref = np.where(host(rows)[:, None] >= 0, feats[np.maximum(host(rows), 0)], 0).astype(np.float32)
    assert np.array_equal(host(g), ref)


# ---------------------------------------------------------------- layers
def _weights(rng, k, cin, cout):
    return (rng.normal(0, 0.3, (k, cin, cout)).astype(np.float32), rng.normal(0, 0.1, cout).astype(np.float32))


WIDTHS = [(16, 16), (64, 64), (32, 128), (48, 80), (128, 32), (16, 256)]    # other model configs: multiples of 16


def test_other_widths_stay_on_the_matrix_cores(rt):
    """the widths a model config may name (ColorModel(config["model"]), codec_pipeline.py:65) are routed to MFMA kernels,
    not to the scalar fallback; odd shapes and the scalar switch still have one"""
    name = lambda op, k, ci, co: rt.lib.pcc_conv_kernel_name(op, k, ci, co).decode()   # noqa: E731
    for ci, co in WIDTHS:
        assert name(0, 27, ci, co) == "k_gconv_gen" and name(0, 8, ci, co) == "k_gconv_gen" and name(1, 27, ci, co) == "k_gconv_gen"
        if co <= 128:
            assert name(2, 8, ci, co) == "k_convT_mfma"
    assert name(0, 27, 32, 32) == "k_gconv16" and name(0, 8, 32, 64) == "k_gconv16" and name(1, 27, 32, 32) == "k_gconv16"
    assert [name(0, 27, 4, co) for co in (16, 32, 64, 128)] == ["k_gconv_first"] * 4
    assert name(0, 27, 3, 5) == "k_gconv_scalar" and name(2, 8, 3, 5) == "k_convT_scalar"
    assert name(2, 8, 32, 32) == "k_convT16" and name(2, 8, 32, 64) == "k_convT_mfma"


@pytest.mark.parametrize("cin,cout", [(4, 32), (32, 32), (32, 64), (3, 5), (32, 1), (4, 16), (4, 64), (4, 128)] + WIDTHS)
@pytest.mark.parametrize("name", ["surf", "tiny", "one"])
@pytest.mark.parametrize("relu", [False, True])
def test_sparse_conv3_bit_exact(rt, oracle, clouds, cin, cout, name, relu):
    rng = np.random.default_rng(cin * 100 + cout)
    keys = sorted_keys(oracle, clouds[name])
    nbr = oracle.map27(keys, 1)
    x = rng.normal(size=(len(keys), cin)).astype(np.float32)
    w, b = _weights(rng, 27, cin, cout)
    out = rt.sparse_conv(dev(rt, x), dev(rt, nbr), dev(rt, w), dev(rt, b), relu)
    ref = oracle.sparse_conv(x, nbr, w, b, relu)
    assert np.array_equal(host(out), ref)


@pytest.mark.parametrize("name", ["surf", "tiny", "one"])
@pytest.mark.parametrize("cin,cout", [(32, 32), (4, 32)] + WIDTHS)
def test_sparse_conv_down_bit_exact(rt, oracle, clouds, cin, cout, name):
    """the stride-2 kernel-2 layers of g_a / h_a"""
    rng = np.random.default_rng(11)
    keys = sorted_keys(oracle, clouds[name])
    pk, nbr8 = oracle.down(keys, 1)
    x = rng.normal(size=(len(keys), cin)).astype(np.float32)
    w, b = _weights(rng, 8, cin, cout)
    for relu in (True, False):
        out = rt.sparse_conv(dev(rt, x), dev(rt, nbr8), dev(rt, w), dev(rt, b), relu)
        assert np.array_equal(host(out), oracle.sparse_conv(x, nbr8, w, b, relu))


@pytest.mark.parametrize("k_vol", [27, 8])
@pytest.mark.parametrize("cout", [32, 64])
def test_conv_rows16_window_edges_and_signed_zero(rt, oracle, k_vol, cout):
    """k_gconv_rows16 (convrows16.h: launches of at most 2048 sixteen-row windows, 1024 for 32 -> 64) on synthetic rule books:
    row counts around a window (1, 15, 16, 17), a ragged last window, the bound itself and one launch past it (k_gconv16),
    rows without any neighbour, whole offsets nobody has, and the two cases a zero operand instead of the select would get
    wrong: a bias of -0.0 on a row that lacks an offset (its accumulator must stay -0.0), and an infinite weight at an
    offset some rows lack (they must not turn into NaN)"""
    rng = np.random.default_rng(1000 * k_vol + cout)
    bound = 2048 * 16 // (cout // 32)
    for n_out in (1, 15, 16, 17, 1000, bound - 5, bound, bound + 1):
        n_in = max(n_out // 2, 3)
        nbr = rng.integers(0, n_in, size=(k_vol, n_out)).astype(np.int32)
        nbr[rng.random((k_vol, n_out)) < 0.6] = -1
        nbr[:, rng.random(n_out) < 0.1] = -1          # rows without a neighbour
        nbr[k_vol // 2, :] = -1                        # an offset nobody has
        x = rng.normal(size=(n_in, 32)).astype(np.float32)
        w, b = _weights(rng, k_vol, 32, cout)
        b[::3] = -0.0
        w[1, 5, 7] = np.inf                            # offset 1 only: rows that lack it stay finite
        for relu in (False, True):
            out = host(rt.sparse_conv(dev(rt, x), dev(rt, nbr), dev(rt, w), dev(rt, b), relu))
            ref = oracle.sparse_conv(x, nbr, w, b, relu)
            nan = np.isnan(ref)                         # inf - inf: a NaN on both sides, whatever its payload
            assert np.array_equal(np.isnan(out), nan), (n_out, relu)
            assert np.array_equal(out.view(np.uint32)[~nan], ref.view(np.uint32)[~nan]), (n_out, relu)   # bits: -0.0 != +0.0 here
            if not relu:
                lacks = nbr[1] < 0
                assert np.isfinite(out[lacks]).all() and (n_out < 100 or not np.isfinite(out[~lacks]).all())
                lone = np.flatnonzero((nbr < 0).all(axis=0))
                assert (out[lone].view(np.uint32)[:, ::3] == 0x80000000).all()     # the bias of -0.0, untouched


def test_conv_linearity(rt, oracle, clouds):
    """size-independent property: conv(a*x) == a*conv(x) for a power of two, zero bias"""
    rng = np.random.default_rng(12)
    keys = sorted_keys(oracle, clouds["surf"])
    nbr = dev(rt, oracle.map27(keys, 1))
    x = rng.normal(size=(len(keys), 32)).astype(np.float32)
    w, _ = _weights(rng, 27, 32, 32)
    b = np.zeros(32, np.float32)
    o1 = host(rt.sparse_conv(dev(rt, x), nbr, dev(rt, w), dev(rt, b), False))
    o2 = host(rt.sparse_conv(dev(rt, x * 4), nbr, dev(rt, w), dev(rt, b), False))
    assert np.array_equal(o1 * 4, o2)


@pytest.mark.parametrize("name", ["surf", "tiny", "one"])
@pytest.mark.parametrize("cin", [32, 4])
def test_sparse_conv_fused_head_bit_exact(rt, oracle, clouds, name, cin):
    """conv3 + ReLU with the 1x1 occupancy head in its epilogue == the two layers of the oracle"""
    rng = np.random.default_rng(77)
    keys = sorted_keys(oracle, clouds[name])
    nbr = oracle.map27(keys, 1)
    x = rng.normal(size=(len(keys), cin)).astype(np.float32)
    w, b = _weights(rng, 27, cin, 32)
    hw = rng.normal(0, 0.3, (32, 1)).astype(np.float32)
    hb = rng.normal(0, 0.1, 1).astype(np.float32)
    feats, logits = rt.sparse_conv_head(dev(rt, x), dev(rt, nbr), dev(rt, w), dev(rt, b), True, dev(rt, hw),
                                        dev(rt, hb))
    ref = oracle.sparse_conv(x, nbr, w, b, True, siblings_first=True)      # the order of g_s's layers (include/pcc.h)
    assert np.array_equal(host(feats), ref)
    assert np.array_equal(host(logits), oracle.linear(ref, hw, hb)[:, 0])


def _structured_cloud(kind, n):
    """coordinate sets that exercise the row-compacting 32->32 kernel (64-row windows, items of 16 slots):
    dense: a full cube, every offset present for the inner rows (four full items per offset);
    dust: isolated voxels, only the centre offset present (single partly filled item, 26 empty offsets);
    children: all 8 children of scattered stride-2 parents (the decoder's candidate sets);
    line: a 1-voxel-wide diagonal, 3 offsets present"""
    rng = np.random.default_rng(n)
    if kind == "dense":
        e = int(round(n ** (1 / 3))) + 1
        g = np.stack(np.meshgrid(np.arange(e), np.arange(e), np.arange(e), indexing="ij"), -1).reshape(-1, 3)[:n]
        pts = g - e // 2
    elif kind == "dust":
        pts = np.unique(rng.integers(-300, 300, (2 * n, 3)) * 3, axis=0)[:n]
    elif kind == "children":
        par = np.unique(rng.integers(-12, 12, (n // 8 + 1, 3)), axis=0)
        offs = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)])
        pts = (par[:, None, :] * 2 + offs[None]).reshape(-1, 3)[:n]
    else:
        t = np.arange(n) - n // 2
        pts = np.stack([t, t, t], 1)
    return np.concatenate([np.zeros((pts.shape[0], 1), np.int64), pts], 1).astype(np.int32)


@pytest.mark.parametrize("cin,cout", [(16, 16), (64, 64), (32, 128), (48, 80)])
@pytest.mark.parametrize("kind", ["dense", "dust", "children", "line"])
@pytest.mark.parametrize("n", [1, 17, 63, 64, 65, 129, 1000])
def test_conv_other_widths_row_compaction_bit_exact(rt, oracle, kind, n, cin, cout):
    """k_gconv_gen (convgen.h): window / item boundaries on neighbourhoods from empty to full for widths other than the
    model default's — chunks of 32 and of 16 input channels, several column blocks, a last block of 16 columns — in
    plain and in siblings-first order (the g_s layers of such a model: conv + pcc_linear head), against the oracle"""
    rng = np.random.default_rng(1000 + n + cin)
    keys = sorted_keys(oracle, _structured_cloud(kind, n))
    nbr = oracle.map27(keys, 1)
    x = rng.normal(size=(len(keys), cin)).astype(np.float32)
    w, b = _weights(rng, 27, cin, cout)
    out = rt.sparse_conv(dev(rt, x), dev(rt, nbr), dev(rt, w), dev(rt, b), False)
    assert np.array_equal(host(out), oracle.sparse_conv(x, nbr, w, b, False))
    hw = rng.normal(0, 0.3, (cout, 1)).astype(np.float32)
    hb = rng.normal(0, 0.1, 1).astype(np.float32)
    feats, logits = rt.sparse_conv_head(dev(rt, x), dev(rt, nbr), dev(rt, w), dev(rt, b), True, dev(rt, hw), dev(rt, hb))
    refr = oracle.sparse_conv(x, nbr, w, b, True, siblings_first=True)
    assert np.array_equal(host(feats), refr)
    assert np.array_equal(host(logits), oracle.linear(refr, hw, hb)[:, 0])
    # registered weights give the same bits
    wd = dev(rt, w)
    rt.conv_prepare(wd)
    assert np.array_equal(host(rt.sparse_conv(dev(rt, x), dev(rt, nbr), wd, dev(rt, b), False)), host(out))
    rt.conv_forget(wd)


@pytest.mark.parametrize("kind", ["dense", "dust", "children", "line"])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 31, 33, 47, 49, 63, 64, 65, 127, 129, 1000, 4099])
def test_conv32_row_compaction_bit_exact(rt, oracle, kind, n):
    """window / item boundaries of k_gconv16 (rows 63|64|65, items 16|17, 32|33, 48|49) on neighbourhoods
    from empty to full, plain and fused-head entry points, against the oracle's fmaf chain"""
    rng = np.random.default_rng(1000 + n)
    keys = sorted_keys(oracle, _structured_cloud(kind, n))
    nbr = oracle.map27(keys, 1)
    x = rng.normal(size=(len(keys), 32)).astype(np.float32)
    w, b = _weights(rng, 27, 32, 32)
    hw = rng.normal(0, 0.3, (32, 1)).astype(np.float32)
    hb = rng.normal(0, 0.1, 1).astype(np.float32)
    ref = oracle.sparse_conv(x, nbr, w, b, False)
    out = rt.sparse_conv(dev(rt, x), dev(rt, nbr), dev(rt, w), dev(rt, b), False)
    assert np.array_equal(host(out), ref)
    feats, logits = rt.sparse_conv_head(dev(rt, x), dev(rt, nbr), dev(rt, w), dev(rt, b), True, dev(rt, hw),
                                        dev(rt, hb))
    refr = oracle.sparse_conv(x, nbr, w, b, True, siblings_first=True)
    assert np.array_equal(host(feats), refr)
    assert np.array_equal(host(logits), oracle.linear(refr, hw, hb)[:, 0])
    # the 32 -> 64 layer (the two column halves as grid.y) on the same neighbourhoods
    w64, b64 = _weights(rng, 27, 32, 64)
    out64 = rt.sparse_conv(dev(rt, x), dev(rt, nbr), dev(rt, w64), dev(rt, b64), True)
    assert np.array_equal(host(out64), oracle.sparse_conv(x, nbr, w64, b64, True))


@pytest.mark.parametrize("kind", ["dense", "dust", "children", "line"])
@pytest.mark.parametrize("n", [1, 7, 8, 9, 15, 16, 17, 31, 33, 63, 500, 2100])
def test_conv_head_up_forms_the_child_rule_book_in_kernel(rt, oracle, kind, n):
    """pcc_sparse_conv_head_up (k_gconv_up: windows of 16 parents, the siblings as a dense product, the other-parent
    neighbours compacted per offset; `dense` clouds fill more than the four pipelined items of an offset): conv on the
    8N generative children given only the parents' rule book, against the oracle's siblings-first conv over the hashed
    child rule book; pcc_subset_map_up: the rule book of a pruned subset of those children against the oracle's hash
    build over the kept keys"""
    rng = np.random.default_rng(77 + n)
    pts = _structured_cloud(kind, n) * 2
    pkeys = sorted_keys(oracle, pts)                       # stride-2 parents
    nbr_p = oracle.map27(pkeys, 2)
    ckeys = oracle.up(pkeys, 2)                            # row 8p + o
    nbr_c = oracle.map27(ckeys, 1)
    x = rng.normal(size=(len(ckeys), 32)).astype(np.float32)
    w, b = _weights(rng, 27, 32, 32)
    hw = rng.normal(0, 0.3, (32, 1)).astype(np.float32)
    hb = rng.normal(0, 0.1, 1).astype(np.float32)
    ref = oracle.sparse_conv(x, nbr_c, w, b, True, siblings_first=True)
    feats, logits = rt.sparse_conv_head_up(dev(rt, x), dev(rt, nbr_p), dev(rt, w), dev(rt, b), True, dev(rt, hw),
                                           dev(rt, hb))
    assert np.array_equal(host(feats), ref)
    assert np.array_equal(host(logits), oracle.linear(ref, hw, hb)[:, 0])
    # a pitch wider than the parent count must be honoured
    wide = np.full((27, len(pkeys) + 5), -1, np.int32)
    wide[:, :len(pkeys)] = nbr_p
    rtm = pkg("runtime")
    xd, wide_d, wd, bd, hwd, hbd = dev(rt, x), dev(rt, wide), dev(rt, w), dev(rt, b), dev(rt, hw), dev(rt, hb)
    f2 = rt.empty((len(ckeys), 32), torch.float32)
    l2 = rt.empty((len(ckeys),), torch.float32)
    rtm.check(rt.lib.pcc_sparse_conv_head_up(rt.ctx, rtm._ptr(xd), len(pkeys), rtm._ptr(wide_d), wide.shape[1],
                                             rtm._ptr(wd), rtm._ptr(bd), 1, rtm._ptr(f2), rtm._ptr(hwd),
                                             rtm._ptr(hbd), rtm._ptr(l2)), "pcc_sparse_conv_head_up")
    assert np.array_equal(host(f2), ref)
    # the explicit child rule book gives the same bits (k_gconv16's two passes over the offsets)
    f3, l3 = rt.sparse_conv_head(xd, dev(rt, nbr_c), wd, bd, True, hwd, hbd)
    assert np.array_equal(host(f3), ref) and np.array_equal(host(l3), host(logits))

    keep = np.sort(rng.choice(len(ckeys), size=max(1, len(ckeys) // 3), replace=False)).astype(np.uint32)
    keep_d = dev(rt, keep.view(np.int32))
    remap = rt.inverse_rows(keep_d, len(ckeys))
    got = host(rt.subset_map_up(dev(rt, nbr_p), keep_d, remap))
    assert np.array_equal(got, oracle.map27(ckeys[keep], 1))


@pytest.mark.parametrize("n_par", [1, 7, 16, 21, 35])
@pytest.mark.parametrize("relu", [True, False])
def test_up_stage_kernels_stay_inside_their_outputs(rt, oracle, n_par, relu):
    """k_gconv_up, k_convT16 and k_convT16p drop the stores of rows past the end through the bounds of their raw buffer
    descriptors (no branch behind them): outputs allocated with a canary tail, parent counts that are not multiples of
    the 16-parent windows / tiles — the tail must be untouched and the rows must equal the oracle's, with and without
    the ReLU (the code paths of the epilogues differ)"""
    rtm = pkg("runtime")
    rng = np.random.default_rng(900 + n_par)
    pts = _structured_cloud("dense", n_par)
    pts[:, 1:] *= 2
    pkeys = sorted_keys(oracle, pts)
    assert len(pkeys) == n_par
    nbr_p = oracle.map27(pkeys, 2)
    ckeys = oracle.up(pkeys, 2)
    nbr_c = oracle.map27(ckeys, 1)
    n = len(ckeys)
    x = rng.normal(size=(n, 32)).astype(np.float32)
    w, b = _weights(rng, 27, 32, 32)
    hw = rng.normal(0, 0.3, (32, 1)).astype(np.float32)
    hb = rng.normal(0, 0.1, 1).astype(np.float32)
    ref = oracle.sparse_conv(x, nbr_c, w, b, relu, siblings_first=True)
    tail = 4096
    canary = 12345.5
    feats = torch.full((n * 32 + tail,), canary, dtype=torch.float32, device="cuda")
    logits = torch.full((n + tail,), canary, dtype=torch.float32, device="cuda")
    xd, nd, wd, bd, hwd, hbd = dev(rt, x), dev(rt, nbr_p), dev(rt, w), dev(rt, b), dev(rt, hw), dev(rt, hb)
    rtm.check(rt.lib.pcc_sparse_conv_head_up(rt.ctx, rtm._ptr(xd), n_par, rtm._ptr(nd), nbr_p.shape[1], rtm._ptr(wd), rtm._ptr(bd),
                                             1 if relu else 0, rtm._ptr(feats), rtm._ptr(hwd), rtm._ptr(hbd), rtm._ptr(logits)),
              "pcc_sparse_conv_head_up")
    f, lg = feats.cpu().numpy(), logits.cpu().numpy()
    assert np.array_equal(f[:n * 32].reshape(n, 32), ref) and np.all(f[n * 32:] == canary)
    assert np.array_equal(lg[:n], oracle.linear(ref, hw, hb)[:, 0]) and np.all(lg[n:] == canary)
    # the up stage (k_convT16 for these sizes; k_convT16p from 4 tiles per resident wave on: the full-size tests)
    w8, b8 = _weights(rng, 8, 32, 32)
    xp = rng.normal(size=(n_par, 32)).astype(np.float32)
    up = torch.full((n * 32 + tail,), canary, dtype=torch.float32, device="cuda")
    xpd, w8d, b8d = dev(rt, xp), dev(rt, w8), dev(rt, b8)
    rtm.check(rt.lib.pcc_convT_gen(rt.ctx, rtm._ptr(xpd), n_par, rtm._ptr(w8d), rtm._ptr(b8d), 32, 32, 1 if relu else 0,
                                   rtm._ptr(up)), "pcc_convT_gen")
    u = up.cpu().numpy()
    assert np.array_equal(u[:n * 32].reshape(n, 32), oracle.convT(xp, w8, b8, relu)) and np.all(u[n * 32:] == canary)


def test_persistent_up_stage_stays_inside_its_output(rt, oracle):
    """k_convT16p (persistent waves, from 4 tiles of 16 parents per resident wave on): a parent count that is not a
    multiple of 16, output with a canary tail, plain and gathered input rows"""
    rtm = pkg("runtime")
    rng = np.random.default_rng(31)
    waves = 2 * 4 * torch.cuda.get_device_properties(0).multi_processor_count
    n_par = 16 * 4 * waves + 16 * 37 + 5
    x = rng.normal(size=(n_par, 32)).astype(np.float32)
    w8, b8 = _weights(rng, 8, 32, 32)
    canary, tail = -777.25, 8192
    ref = oracle.convT(x, w8, b8, True)
    xd, w8d, b8d = dev(rt, x), dev(rt, w8), dev(rt, b8)
    up = torch.full((8 * n_par * 32 + tail,), canary, dtype=torch.float32, device="cuda")
    rtm.check(rt.lib.pcc_convT_gen(rt.ctx, rtm._ptr(xd), n_par, rtm._ptr(w8d), rtm._ptr(b8d), 32, 32, 1, rtm._ptr(up)), "pcc_convT_gen")
    u = up.cpu().numpy()
    assert np.array_equal(u[:8 * n_par * 32].reshape(-1, 32), ref) and np.all(u[8 * n_par * 32:] == canary)
    rows = rng.permutation(n_par).astype(np.uint32)
    up.fill_(canary)
    rd = dev(rt, rows.view(np.int32))
    rtm.check(rt.lib.pcc_convT_gen_gather(rt.ctx, rtm._ptr(xd), rtm._ptr(rd), n_par, rtm._ptr(w8d), rtm._ptr(b8d), 1, rtm._ptr(up)),
              "pcc_convT_gen_gather")
    u = up.cpu().numpy()
    assert np.array_equal(u[:8 * n_par * 32].reshape(-1, 32), oracle.convT(x[rows], w8, b8, True)) and np.all(u[8 * n_par * 32:] == canary)


@pytest.mark.parametrize("kind", ["dense", "dust"])
def test_conv32_large_launch_bit_exact(rt, oracle, kind):
    """launches of 100k rows and more run on 64-row windows (smaller ones on 32-row windows: the tests above): the same
    entry points just past that size, a ragged last window included"""
    rng = np.random.default_rng(4242)
    n = 200_000 + 77
    keys = sorted_keys(oracle, _structured_cloud(kind, n))
    assert len(keys) >= 200_000
    nbr = oracle.map27(keys, 1)
    x = rng.normal(size=(len(keys), 32)).astype(np.float32)
    w, b = _weights(rng, 27, 32, 32)
    hw = rng.normal(0, 0.3, (32, 1)).astype(np.float32)
    hb = rng.normal(0, 0.1, 1).astype(np.float32)
    ref = oracle.sparse_conv(x, nbr, w, b, True, siblings_first=True)
    feats, logits = rt.sparse_conv_head(dev(rt, x), dev(rt, nbr), dev(rt, w), dev(rt, b), True, dev(rt, hw), dev(rt, hb))
    assert np.array_equal(host(feats), ref)
    assert np.array_equal(host(logits), oracle.linear(ref, hw, hb)[:, 0])
    w64, b64 = _weights(rng, 27, 32, 64)
    out64 = rt.sparse_conv(dev(rt, x), dev(rt, nbr), dev(rt, w64), dev(rt, b64), False)
    assert np.array_equal(host(out64), oracle.sparse_conv(x, nbr, w64, b64, False))
    # the in-kernel child rule book on 8 x 25_010 candidates
    pkeys = sorted_keys(oracle, _structured_cloud(kind, 25_010) * 2)
    nbr_p = oracle.map27(pkeys, 2)
    ckeys = oracle.up(pkeys, 2)
    assert len(ckeys) >= 200_000
    xc = rng.normal(size=(len(ckeys), 32)).astype(np.float32)
    refc = oracle.sparse_conv(xc, oracle.map27(ckeys, 1), w, b, True, siblings_first=True)
    fc, lc = rt.sparse_conv_head_up(dev(rt, xc), dev(rt, nbr_p), dev(rt, w), dev(rt, b), True, dev(rt, hw), dev(rt, hb))
    assert np.array_equal(host(fc), refc)
    assert np.array_equal(host(lc), oracle.linear(refc, hw, hb)[:, 0])


def test_conv_wide_row_form_is_bit_exact():
    """Launches whose input tensor has 2^25 rows or more (4 GB), or whose parent rule book has a pitch of 2^24 or more,
    run k_gconv16 with 64-bit row arithmetic instead of 32-bit byte offsets in the slot records (conv16.h, WIDE).
    PCC_CONV_WIDE_ROWS=1 (read once per process) sends every launch that way: the convolution tests of this file once
    more under it, in a child process, against the same oracle results"""
    import subprocess
    import sys
    env = dict(os.environ, PCC_CONV_WIDE_ROWS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "conv and not switch and not wide_row and not legacy and not rows16"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-1500:], r.stderr[-500:])


def test_conv_small_launches_rows16_switch_is_bit_exact():
    """Launches of at most 2048 waves on an explicit rule book run k_gconv_rows16 (convrows16.h: 16-row windows, no
    compaction); PCC_CONV_ROWS16_MAX=0 (read once per process) keeps them on k_gconv16's 32-row windows, the form they
    took before: the convolution tests of this file once more under it, in a child process, against the same oracle
    results"""
    import subprocess
    import sys
    env = dict(os.environ, PCC_CONV_ROWS16_MAX="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "conv and not switch and not wide_row and not legacy and not rows16"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-1500:], r.stderr[-500:])


def test_conv_head_up_legacy_form_is_bit_exact():
    """PCC_CONV_UP_LEGACY=1 (read once per process) keeps the g_s layers on k_gconv16's in-kernel-rule-book form (two
    passes over the offsets: siblings, then the rest) instead of k_gconv_up — the form tensors of 2^25 rows and more
    take: the head-up tests of this file once more under it, in a child process, against the same oracle results"""
    import subprocess
    import sys
    env = dict(os.environ, PCC_CONV_UP_LEGACY="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "head_up_forms or large_launch"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-1500:], r.stderr[-500:])


def test_conv_head_up_refuses_under_the_scalar_switch(rt):
    """the form with the in-kernel rule book exists only as an MFMA kernel: under PCC_FORCE_SCALAR=1 it must be
    refused (the decoder then materialises the child rule books), not silently computed by something else"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, importlib, torch; sys.path.insert(0, %r)\n"
            "p = importlib.import_module('demo-learned-point-cloud-compression_amd')\n"
            "rtm = importlib.import_module('demo-learned-point-cloud-compression_amd.runtime')\n"
            "rt = rtm.Runtime()\n"
            "x = torch.zeros((8, 32), device='cuda'); nb = torch.full((27, 1), -1, dtype=torch.int32, device='cuda')\n"
            "w = torch.zeros((27, 32, 32), device='cuda'); b = torch.zeros(32, device='cuda')\n"
            "hw = torch.zeros((32, 1), device='cuda'); hb = torch.zeros(1, device='cuda')\n"
            "try:\n    rt.sparse_conv_head_up(x, nb, w, b, True, hw, hb)\n"
            "except rtm.PccError as e:\n    print('REFUSED', e.code)\nelse:\n    print('RAN')\n") % root
    for env, want in (({"PCC_FORCE_SCALAR": "1"}, "REFUSED"), ({}, "RAN")):
        out = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True,
                             timeout=300)
        assert want in out.stdout, (env, out.stdout, out.stderr[-400:])


def test_prepared_weights_give_the_same_bits(rt, oracle):
    """pcc_conv_prepare: a registered weight tensor is re-arranged once and found by its pointer; the results equal
    those of the per-call path and the oracle, a refresh after the contents changed is honoured, and forgetting
    the pointer falls back to the per-call copy"""
    rng = np.random.default_rng(77)
    keys = sorted_keys(oracle, _structured_cloud("children", 700))
    nbr = oracle.map27(keys, 1)
    x = rng.normal(size=(len(keys), 32)).astype(np.float32)
    for cout in (32, 64):
        w, b = _weights(rng, 27, 32, cout)
        xd, nd, wd, bd = dev(rt, x), dev(rt, nbr), dev(rt, w), dev(rt, b)
        ref = oracle.sparse_conv(x, nbr, w, b, True)
        assert np.array_equal(host(rt.sparse_conv(xd, nd, wd, bd, True)), ref)
        rt.conv_prepare(wd)
        assert np.array_equal(host(rt.sparse_conv(xd, nd, wd, bd, True)), ref)
        w2 = (w * np.float32(0.5)).astype(np.float32)
        wd.copy_(dev(rt, w2))                                  # same pointer, new contents
        rt.conv_prepare(wd)
        assert np.array_equal(host(rt.sparse_conv(xd, nd, wd, bd, True)), oracle.sparse_conv(x, nbr, w2, b, True))
        rt.conv_forget(wd)
        assert np.array_equal(host(rt.sparse_conv(xd, nd, wd, bd, True)), oracle.sparse_conv(x, nbr, w2, b, True))
    with pytest.raises(pkg("runtime").PccError):
        rt.conv_prepare(dev(rt, np.zeros((27, 4, 32), np.float32)))   # the 4 -> 32 layer has no pre-arranged form


@pytest.mark.parametrize("cout", [32, 64])
@pytest.mark.parametrize("n,m", [(5000, 1300), (40, 40), (3000, 1)])
def test_conv_on_a_subset_of_the_rows(rt, oracle, cout, n, m):
    """a conv whose output is only sampled afterwards, evaluated at the sampled rows alone (pcc_gather_map_columns +
    pcc_sparse_conv with n_out != n_in): the same bits as the rows of the full conv, absent rows (-1) zeroed by
    pcc_gather_rows_or_zero — the native codec's form of h_s + features_at_coordinates"""
    rng = np.random.default_rng(n + cout)
    keys = sorted_keys(oracle, _structured_cloud("children", n))
    n = len(keys)
    nbr = oracle.map27(keys, 1)
    x = rng.normal(size=(n, 32)).astype(np.float32)
    w, b = _weights(rng, 27, 32, cout)
    full = oracle.sparse_conv(x, nbr, w, b, False)
    rows = rng.choice(n, size=min(m, n), replace=False).astype(np.int32)
    rows[::5] = -1                                         # coordinates that are not in the set
    sub, me = rt.gather_map_columns(dev(rt, nbr), dev(rt, rows))
    assert np.array_equal(host(sub), np.where(rows[None, :] >= 0, nbr[:, np.maximum(rows, 0)], -1))
    out = rt.sparse_conv(dev(rt, x), sub, dev(rt, w), dev(rt, b), False)
    got = host(rt.gather_rows_or_zero(out, me))
    ref = np.where(rows[:, None] >= 0, full[np.maximum(rows, 0)], 0).astype(np.float32)
    assert np.array_equal(got, ref)


def test_conv32_rule_book_with_pitch_and_foreign_input(rt, oracle):
    """n_in != n_out (stride-2 conv: 8 offsets, input = children, output = parents) and a rule book whose
    row pitch is larger than n_out: the kernel must honour the pitch and never read past n_out"""
    rng = np.random.default_rng(5)
    keys = sorted_keys(oracle, _structured_cloud("children", 3000))
    pk, nbr8 = oracle.down(keys, 1)
    x = rng.normal(size=(len(keys), 32)).astype(np.float32)
    w, b = _weights(rng, 8, 32, 32)
    ref = oracle.sparse_conv(x, nbr8, w, b, True)
    wide = np.full((8, nbr8.shape[1] + 37), -1, np.int32)
    wide[:, :nbr8.shape[1]] = nbr8
    rtm = pkg("runtime")
    xd, wide_d, wd, bd = dev(rt, x), dev(rt, wide), dev(rt, w), dev(rt, b)
    n_out = nbr8.shape[1]
    out = rt.empty((n_out, 32), torch.float32)
    rtm.check(rt.lib.pcc_sparse_conv(rt.ctx, rtm._ptr(xd), x.shape[0], rtm._ptr(wide_d), 8, wide.shape[1], n_out,
                                     rtm._ptr(wd), rtm._ptr(bd), 32, 32, 1, rtm._ptr(out)), "pcc_sparse_conv")
    assert np.array_equal(host(out), ref)


@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (5, 3), (16, 16), (64, 64), (32, 128), (48, 80), (128, 32)])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 31, 32, 33, 700])
def test_convT_gen_bit_exact(rt, oracle, cin, cout, n):
    rng = np.random.default_rng(n)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    w, b = _weights(rng, 8, cin, cout)
    out = rt.convT_gen(dev(rt, x), dev(rt, w), dev(rt, b), True)
    assert np.array_equal(host(out), oracle.convT(x, w, b, True))


@pytest.mark.parametrize("cin,cout", [(32, 1), (32, 3), (7, 9)])
def test_linear_bit_exact(rt, oracle, cin, cout):
    rng = np.random.default_rng(cin)
    x = rng.normal(size=(1000, cin)).astype(np.float32)
    w = rng.normal(0, 0.3, (cin, cout)).astype(np.float32)
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    out = rt.linear(dev(rt, x), dev(rt, w), dev(rt, b), False)
    assert np.array_equal(host(out), oracle.linear(x, w, b))


def test_convT_on_gathered_rows(rt, oracle):
    """the up stage on the kept rows in place (pcc_convT_gen_gather) == gather then convT, == oracle"""
    rng = np.random.default_rng(10)
    x = rng.normal(size=(3000, 32)).astype(np.float32)
    rows = np.sort(rng.choice(3000, 1001, replace=False)).astype(np.int32)
    w, b = _weights(rng, 8, 32, 32)
    out = rt.convT_gen_gather(dev(rt, x), dev(rt, rows), dev(rt, w), dev(rt, b), True)
    assert np.array_equal(host(out), oracle.convT(x[rows], w, b, True))


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("n", [131072, 140001, 170017])
def test_convT_persistent_waves_bit_exact(rt, oracle, n, relu):
    """from 4 tiles per resident wave on, the 32 -> 32 up stage runs as persistent waves (k_convT16p: weights in registers,
    tiles walked with a stride of the grid): whole tiles, a ragged last tile, waves with 4, 5 and 6 tiles; plain and gathered"""
    rng = np.random.default_rng(n)
    x = rng.normal(size=(n + 1000, 32)).astype(np.float32)
    w, b = _weights(rng, 8, 32, 32)
    out = rt.convT_gen(dev(rt, x[:n]), dev(rt, w), dev(rt, b), relu)
    assert np.array_equal(host(out), oracle.convT(x[:n], w, b, relu))
    rows = np.sort(rng.choice(n + 1000, n, replace=False)).astype(np.int32)
    out = rt.convT_gen_gather(dev(rt, x), dev(rt, rows), dev(rt, w), dev(rt, b), relu)
    assert np.array_equal(host(out), oracle.convT(x[rows], w, b, relu))


def test_linear_on_gathered_rows(rt, oracle):
    """the colour head on the kept rows in place (pcc_linear_gather) == gather then linear, == oracle"""
    rng = np.random.default_rng(9)
    x = rng.normal(size=(5000, 32)).astype(np.float32)
    rows = np.sort(rng.choice(5000, 1777, replace=False)).astype(np.int32)
    for cout in (3, 1, 8):
        w = rng.normal(0, 0.3, (32, cout)).astype(np.float32)
        b = rng.normal(0, 0.1, cout).astype(np.float32)
        out = rt.linear_gather(dev(rt, x), dev(rt, rows), dev(rt, w), dev(rt, b), False)
        assert np.array_equal(host(out), oracle.linear(x[rows], w, b))


# ---------------------------------------------------------------- top-k
@pytest.mark.parametrize("case", ["random", "ties", "all_equal", "k_zero_and_all", "negatives"])
def test_topk_prune(rt, oracle, case):
    rng = np.random.default_rng({"random": 1, "ties": 2, "all_equal": 3, "k_zero_and_all": 4, "negatives": 5}[case])
    counts = [3000, 1, 0, 5000]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    n = offs[-1]
    if case == "random":
        lg = rng.normal(size=n).astype(np.float32)
        k = [1000, 1, 0, 2500]
    elif case == "ties":
        lg = rng.integers(-3, 3, n).astype(np.float32)  # heavy ties at the threshold
        k = [1234, 1, 0, 4999]
    elif case == "all_equal":
        lg = np.full(n, 0.25, np.float32)
        k = [17, 1, 0, 100]
    elif case == "k_zero_and_all":
        lg = rng.normal(size=n).astype(np.float32)
        k = [0, 5, 0, 5000]
    else:
        lg = -np.abs(rng.normal(size=n)).astype(np.float32)
        lg[::7] = 0.0
        lg[::13] = -0.0
        k = [500, 0, 0, 700]
    kc = [min(a, b) for a, b in zip(k, counts)]
    keep = rt.topk_prune(dev(rt, lg), offs, kc)
    ref = oracle.topk(lg, offs, kc)
    assert np.array_equal(host(keep).view(np.uint32), ref)
    # the same with the rows' positions among the kept ones (-1: not kept) written by the placement
    keep2, remap = rt.topk_prune(dev(rt, lg), offs, kc, with_map=True)
    want = np.full(n, -1, np.int32)
    want[ref.astype(np.int64)] = np.arange(len(ref), dtype=np.int32)
    assert np.array_equal(host(keep2).view(np.uint32), ref) and np.array_equal(host(remap), want)


@pytest.mark.parametrize("case", ["large", "large_ties", "twelve_frames"])
def test_topk_prune_many_blocks(rt, oracle, case):
    """frames of up to 2M candidates (hundreds of histogram blocks per frame: the last block of a frame to finish
    closes the pass), ties spread over a whole frame, and a 12-frame GOP (parameters staged through pinned
    memory instead of kernel arguments)"""
    rng = np.random.default_rng({"large": 31, "large_ties": 32, "twelve_frames": 33}[case])
    if case == "twelve_frames":
        counts = [int(c) for c in rng.integers(0, 40000, 12)]
        counts[3] = 0
    else:
        counts = [700_000, 1, 2_000_000, 300]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    n = offs[-1]
    if case == "large_ties":
        lg = (rng.integers(-50, 50, n) / 8).astype(np.float32)
    else:
        lg = rng.normal(size=n).astype(np.float32)
    k = [int(c * f) for c, f in zip(counts, rng.random(len(counts)))]
    keep = rt.topk_prune(dev(rt, lg), offs, k)
    ref = oracle.topk(lg, offs, k)
    assert np.array_equal(host(keep).view(np.uint32), ref)
    keep2, remap = rt.topk_prune(dev(rt, lg), offs, k, with_map=True)
    want = np.full(n, -1, np.int32)
    want[ref.astype(np.int64)] = np.arange(len(ref), dtype=np.int32)
    assert np.array_equal(host(keep2).view(np.uint32), ref) and np.array_equal(host(remap), want)


# ---------------------------------------------------------------- entropy kernels
def test_factorized_quant_dequant(rt, oracle):
    rng = np.random.default_rng(21)
    z = (rng.normal(size=(777, 32)) * 3).astype(np.float32)
    z[0, :] = np.float32(0.5) + oracle.t["entropy_bottleneck.medians"]  # exact ties -> round half even
    med = dev(rt, oracle.t["entropy_bottleneck.medians"])
    sym, zhat = rt.factorized_quant(dev(rt, z), med)
    rs, rz = oracle.factorized_quant(z)
    assert np.array_equal(host(sym), rs) and np.array_equal(host(zhat), rz)
    assert np.array_equal(host(rt.factorized_dequant(sym, med)), oracle.factorized_dequant(rs))


def test_gaussian_quant_indexes_dequant(rt, oracle):
    rng = np.random.default_rng(22)
    n, c = 1501, 32
    y = (rng.normal(size=(n, c)) * 2).astype(np.float32)
    params = np.concatenate([np.abs(rng.normal(1.5, 3.0, (n, c))), rng.normal(0, 1, (n, c))], 1).astype(np.float32)
    params[:5, :c] = -1.0      # below the lower bound
    params[5:10, :c] = 1000.0  # above the table
    scale = (oracle.scale_nn([[1, 0], [0, 1], [1, 1]]) + oracle.eps).astype(np.float32)
    tab = dev(rt, oracle.t["gaussian_conditional.scale_table"])
    sym, idx = rt.gaussian_quant(dev(rt, y), dev(rt, params), dev(rt, scale), tab)
    rs, ri = oracle.gaussian_quant(y, params, scale)
    assert np.array_equal(host(sym), rs) and np.array_equal(host(idx), ri)
    s16, i8, flag = rt.gaussian_quant16(dev(rt, y), dev(rt, params), dev(rt, scale), tab)
    assert int(flag.item()) == 0
    assert np.array_equal(host(s16).astype(np.int32), rs) and np.array_equal(host(i8).astype(np.int32), ri)
    _, _, flag2 = rt.gaussian_quant16(dev(rt, y * 1e5), dev(rt, params), dev(rt, scale), tab)
    assert int(flag2.item()) == 1      # symbols beyond int16 are flagged, the caller falls back to int32
    assert np.array_equal(host(rt.gaussian_indexes8(dev(rt, params), dev(rt, scale[2]), tab)).astype(np.int32),
                          ri[2])
    i1 = rt.gaussian_indexes(dev(rt, params), dev(rt, scale[2]), tab)
    assert np.array_equal(host(i1), oracle.gaussian_indexes(params, scale[2]))
    assert np.array_equal(host(i1), ri[2])
    yh = rt.gaussian_dequant(dev(rt, rs[2]), dev(rt, params), dev(rt, scale[2]), float(oracle.t[
        "gaussian_conditional.scale_table"][0]), float(oracle.off_a), float(oracle.off_b))
    assert np.array_equal(host(yh), oracle.gaussian_dequant(rs[2], params, scale[2]))


def test_gaussian_indexes_keep_the_counting_definition(rt, oracle):
    """build_indexes counts table entries >= the scale (compressai entropy_models.py GaussianConditional.build_indexes).
    The kernels take a lower-bound search when the table ascends: exact ties, NaN scales and a table that does NOT ascend
    (the count, not the search, is the definition) must give the oracle's indexes."""
    rng = np.random.default_rng(23)
    n, c = 333, 32
    table = np.ascontiguousarray(oracle.t["gaussian_conditional.scale_table"], dtype=np.float32)
    params = np.concatenate([np.abs(rng.normal(1.5, 3.0, (n, c))), rng.normal(0, 1, (n, c))], 1).astype(np.float32)
    one = np.ones(c, dtype=np.float32)
    params[:64, 0] = table           # ties with every table entry (scale 1.0 keeps them exact)
    params[64:70, 1] = np.nan
    params[70:72, 2] = np.inf
    y = rng.normal(size=(n, c)).astype(np.float32)
    saved = oracle.t["gaussian_conditional.scale_table"]
    try:
        for tab_h in (table, table[::-1].copy(), rng.permutation(table)):
            oracle.t["gaussian_conditional.scale_table"] = tab_h
            tab = dev(rt, tab_h)
            want = oracle.gaussian_indexes(params, one)
            assert np.array_equal(host(rt.gaussian_indexes(dev(rt, params), dev(rt, one), tab)), want)
            assert np.array_equal(host(rt.gaussian_indexes8(dev(rt, params), dev(rt, one), tab)).astype(np.int32), want)
            _, idx = rt.gaussian_quant(dev(rt, y), dev(rt, params), dev(rt, one[None, :]), tab)
            assert np.array_equal(host(idx)[0], want)
    finally:
        oracle.t["gaussian_conditional.scale_table"] = saved


# ---------------------------------------------------------------- octree
@pytest.mark.parametrize("name", ["surf", "tiny", "one", "rand"])
def test_octree_blob_matches_oracle_and_round_trips(rt, oracle, clouds, name):
    runtime, utils = pkg("runtime"), pkg("utils")
    c = clouds[name].copy()
    c = c[c[:, 0] == 0]
    c[:, 1:] *= 8
    keys = sorted_keys(oracle, c)
    kd = dev(rt, keys.view(np.int64))
    blob = utils.gpcc_encode(kd, keys.view(np.int64), 0, len(keys), 9)
    ref = oracle.octree_encode(c[:, 1:] // 8, 4096)
    assert blob == ref
    pts = utils.gpcc_decode(blob, 8)
    assert np.array_equal(pts, oracle.keys_to_coords(keys)[:, 1:])
    assert np.array_equal(oracle.octree_decode(blob) * 8, pts)


@pytest.mark.parametrize("n,extent", [(60000, 6000), (65536, 48), (65537, 48), (2, 3), (9, 2)])
def test_octree_single_launch_sizes(rt, oracle, n, extent):
    """the one-workgroup octree kernel: scattered leaves whose occupancy bytes exceed its LDS buffer (60000 leaves
    over +-3000: the bytes are ORed into HBM instead), the largest single-launch input and the first size that
    takes the per-level kernels, and the smallest trees"""
    runtime, utils = pkg("runtime"), pkg("utils")
    rng = np.random.default_rng(n)
    if extent ** 3 < 4 * n:
        g = np.stack(np.meshgrid(*[np.arange(extent)] * 3, indexing="ij"), -1).reshape(-1, 3)
        g = g[rng.permutation(len(g))[:n]] - extent // 2
    else:
        g = np.unique(rng.integers(-extent // 2, extent // 2, (2 * n, 3)), axis=0)
        g = g[rng.permutation(len(g))[:n]]
    c = np.concatenate([np.zeros((len(g), 1), np.int64), g], 1).astype(np.int32)
    keys = sorted_keys(oracle, c)
    blob = utils.gpcc_encode(dev(rt, keys.view(np.int64)), keys.view(np.int64), 0, len(keys), 0)
    assert blob == oracle.octree_encode(c[:, 1:], 32768)
    assert np.array_equal(utils.gpcc_decode(blob, 1), oracle.keys_to_coords(keys)[:, 1:])


def test_octree_stride1_large_extent(rt, oracle):
    """geometry-only use (KITTI-like): stride-1 keys, coordinates spanning +-2000"""
    runtime, utils = pkg("runtime"), pkg("utils")
    rng = np.random.default_rng(9)
    c = random_cloud(rng, 3000, extent=4000, batches=1, lo=-2000)
    keys = sorted_keys(oracle, c)
    blob = utils.gpcc_encode(dev(rt, keys.view(np.int64)), keys.view(np.int64), 0, len(keys), 0)
    assert blob == oracle.octree_encode(c[:, 1:], 32768)
    assert np.array_equal(utils.gpcc_decode(blob, 1), oracle.keys_to_coords(keys)[:, 1:])
