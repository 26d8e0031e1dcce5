"""Integer CDF tables of the two entropy models, built from their raw parameters at load time — what
`model.update()` does in the reference (`compression_model.update()`, sender/encoder/codec_pipeline.py:69,
receiver/decoder/codec_parallel.py:60: CompressAI's EntropyBottleneck.update / GaussianConditional.update over
`pmf_to_quantized_cdf`).  CompressAI 1.2.4 is not in the reference tree: the algorithms are restated from its
published sources ([RECALL]) — compressai/entropy_models/entropy_models.py and
compressai/cpp_exts/ops/ops.cpp.  Host numpy, float32 where CompressAI computes in float32; runs once per model.

Raw parameters a checkpoint (weights.npz) may carry instead of the integer tables:
  entropy_bottleneck._matrix{i} [C, f(i+1), f(i)], ._bias{i} [C, f(i+1), 1], ._factor{i} [C, f(i+1), 1] (i < len(filters)),
  entropy_bottleneck.quantiles [C, 1, 3]          CompressAI's state_dict names; filters from the shapes
  gaussian_conditional.scale_table [L]             (absent: CompressAI's default table, 64 levels 0.11 .. 256)
"""
import numpy as np

PRECISION = 16
TAIL_MASS = 1e-9
SCALE_MIN, SCALE_MAX, SCALE_LEVELS = 0.11, 256.0, 64

EB = "entropy_bottleneck"
GC = "gaussian_conditional"
EB_TABLES = (EB + ".medians", EB + ".quantized_cdf", EB + ".cdf_length", EB + ".offset")
GC_TABLES = (GC + ".quantized_cdf", GC + ".cdf_length", GC + ".offset")


def pmf_to_quantized_cdf(pmf, precision=PRECISION):
    """[RECALL] ops.cpp pmf_to_quantized_cdf: round(p * 2^precision) (half away from zero), rescale to the full range,
    prefix sum, then for every empty bin steal one count from the bin with the smallest frequency > 1 (the first such
    bin; entries between it and the empty bin shift by one)."""
    pmf = np.asarray(pmf, dtype=np.float32)
    if not (np.all(np.isfinite(pmf)) and np.all(pmf >= 0)):
        raise ValueError("pmf_to_quantized_cdf: pmf must be finite and non-negative")
    cdf = np.zeros(pmf.shape[0] + 1, dtype=np.uint64)
    # std::round of the float32 product, half away from zero: the + 0.5 in double (exact) — in float32 the value just below
    # 0.5 plus 0.5 rounds up to 1.0, where std::round gives 0
    cdf[1:] = np.floor((pmf * np.float32(1 << precision)).astype(np.float64) + 0.5).astype(np.uint64)
    total = int(cdf.sum())
    if total <= 0:
        raise ValueError("pmf_to_quantized_cdf: empty pmf")
    cdf = (np.uint64(1 << precision) * cdf) // np.uint64(total)
    cdf = np.cumsum(cdf).astype(np.int64)
    cdf[-1] = 1 << precision
    for i in range(cdf.shape[0] - 1):
        if cdf[i] == cdf[i + 1]:
            freq = np.diff(cdf)
            cand = np.where(freq > 1, freq, np.iinfo(np.int64).max)
            best = int(np.argmin(cand))
            if cand[best] == np.iinfo(np.int64).max:
                raise ValueError("pmf_to_quantized_cdf: no frequency left to steal from")
            if best < i:
                cdf[best + 1:i + 1] -= 1
            else:
                cdf[i + 1:best + 1] += 1
    if not (np.all(np.diff(cdf) > 0) and cdf[0] == 0 and cdf[-1] == (1 << precision)):
        raise ValueError("pmf_to_quantized_cdf: table is not strictly increasing")
    return cdf.astype(np.int32)


def default_scale_table(lo=SCALE_MIN, hi=SCALE_MAX, levels=SCALE_LEVELS):
    """[RECALL] compressai.models.utils / get_scale_table: exp(linspace(log lo, log hi, levels))"""
    return np.exp(np.linspace(np.log(lo), np.log(hi), levels)).astype(np.float32)


def gaussian_tables(scale_table, tail_mass=TAIL_MASS):
    """[RECALL] GaussianConditional.update(): per table row a zero-mean Gaussian of that scale sampled on the integers,
    centre = ceil(scale * -ppf(tail_mass / 2)), the mass beyond the support as the overflow bin.
    -> (quantized_cdf int32 [L, max_len + 2], cdf_length int32 [L], offset int32 [L])"""
    from scipy.special import erfc
    from scipy.stats import norm
    table = np.asarray(scale_table, dtype=np.float32)
    multiplier = np.float32(-norm.ppf(tail_mass / 2))
    center = np.ceil(table * multiplier).astype(np.int32)
    length = 2 * center + 1
    max_len = int(length.max())
    samples = np.abs(np.arange(max_len, dtype=np.int32)[None, :] - center[:, None]).astype(np.float32)
    scale = table[:, None]

    def phi(x):   # _standardized_cumulative: 0.5 erfc(-x / sqrt 2)
        return (np.float32(0.5) * erfc(np.float32(-(2 ** -0.5)) * x.astype(np.float32))).astype(np.float32)

    upper = phi((np.float32(0.5) - samples) / scale)
    lower = phi((np.float32(-0.5) - samples) / scale)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdfs = np.zeros((table.shape[0], max_len + 2), dtype=np.int32)
    for i in range(table.shape[0]):
        c = pmf_to_quantized_cdf(np.concatenate([pmf[i, :length[i]], tail[i]]))
        cdfs[i, :c.shape[0]] = c
    return cdfs, (length + 2).astype(np.int32), (-center).astype(np.int32)


def _softplus(x):
    x = np.asarray(x, dtype=np.float32)
    return np.where(x > np.float32(20), x, np.log1p(np.exp(np.minimum(x, np.float32(20))))).astype(np.float32)


def _sigmoid(x):
    return (np.float32(1) / (np.float32(1) + np.exp(-np.asarray(x, dtype=np.float32)))).astype(np.float32)


def bottleneck_raw(tensors):
    """the raw EntropyBottleneck parameters of a checkpoint as (matrices, biases, factors, quantiles), or None"""
    mats, biases, factors = [], [], []
    i = 0
    while f"{EB}._matrix{i}" in tensors:
        mats.append(np.asarray(tensors[f"{EB}._matrix{i}"], dtype=np.float32))
        biases.append(np.asarray(tensors[f"{EB}._bias{i}"], dtype=np.float32))
        if f"{EB}._factor{i}" in tensors:
            factors.append(np.asarray(tensors[f"{EB}._factor{i}"], dtype=np.float32))
        i += 1
    if not mats or f"{EB}.quantiles" not in tensors:
        return None
    if len(factors) != len(mats) - 1:
        raise ValueError(f"{EB}: {len(mats)} matrices need {len(mats) - 1} factors, found {len(factors)}")
    return mats, biases, factors, np.asarray(tensors[f"{EB}.quantiles"], dtype=np.float32)


def logits_cumulative(x, mats, biases, factors):
    """[RECALL] EntropyBottleneck._logits_cumulative on x [C, 1, N]: per channel a small monotone MLP,
    logits = softplus(M_i) @ logits + b_i, then (all but the last layer) logits += tanh(f_i) * tanh(logits)"""
    logits = np.asarray(x, dtype=np.float32)
    for i, (m, b) in enumerate(zip(mats, biases)):
        logits = (np.matmul(_softplus(m), logits) + b).astype(np.float32)
        if i < len(factors):
            logits = (logits + np.tanh(factors[i]) * np.tanh(logits)).astype(np.float32)
    return logits


def bottleneck_tables(mats, biases, factors, quantiles):
    """[RECALL] EntropyBottleneck.update(): medians = quantiles[:, 0, 1]; support [median - ceil(median - q0),
    median + ceil(q2 - median)]; pmf from sigmoid differences of the cumulative logits at +- 0.5; the mass outside the
    sampled range (left of the first sample, right of the LAST sample of the longest row) as the overflow bin.
    -> (medians f32 [C], quantized_cdf int32 [C, max_len + 2], cdf_length int32 [C], offset int32 [C])"""
    q = np.asarray(quantiles, dtype=np.float32)
    medians = q[:, 0, 1].copy()
    minima = np.maximum(np.ceil(medians - q[:, 0, 0]), 0).astype(np.int32)
    maxima = np.maximum(np.ceil(q[:, 0, 2] - medians), 0).astype(np.int32)
    pmf_start = medians - minima.astype(np.float32)
    pmf_length = maxima + minima + 1
    max_len = int(pmf_length.max())
    samples = (pmf_start[:, None, None] + np.arange(max_len, dtype=np.float32)[None, None, :]).astype(np.float32)
    lower = logits_cumulative(samples - np.float32(0.5), mats, biases, factors)
    upper = logits_cumulative(samples + np.float32(0.5), mats, biases, factors)
    sign = -np.sign(lower + upper)
    pmf = np.abs(_sigmoid(sign * upper) - _sigmoid(sign * lower))[:, 0, :]
    tail = (_sigmoid(lower[:, 0, :1]) + _sigmoid(-upper[:, 0, -1:])).astype(np.float32)
    cdfs = np.zeros((q.shape[0], max_len + 2), dtype=np.int32)
    for c in range(q.shape[0]):
        t = pmf_to_quantized_cdf(np.concatenate([pmf[c, :pmf_length[c]], tail[c]]))
        cdfs[c, :t.shape[0]] = t
    return medians.astype(np.float32), cdfs, (pmf_length + 2).astype(np.int32), (-minima).astype(np.int32)


def update_tensors(tensors, force=False):
    """model.update(): (re)build the integer tables a checkpoint lacks (all of them with force=True) from its raw
    parameters, in place.  Returns True when a table was built (CompressAI's `updated` flag)."""
    updated = False
    if force or any(k not in tensors for k in GC_TABLES):
        if GC + ".scale_table" not in tensors:
            tensors[GC + ".scale_table"] = default_scale_table()
        cdf, length, offset = gaussian_tables(tensors[GC + ".scale_table"])
        tensors[GC + ".quantized_cdf"], tensors[GC + ".cdf_length"], tensors[GC + ".offset"] = cdf, length, offset
        updated = True
    if force or any(k not in tensors for k in EB_TABLES):
        raw = bottleneck_raw(tensors)
        if raw is None:
            if any(k not in tensors for k in EB_TABLES):
                raise KeyError(f"checkpoint has neither the integer tables of {EB} nor its raw parameters "
                               f"({EB}._matrix0 ..., {EB}.quantiles)")
        else:
            med, cdf, length, offset = bottleneck_tables(*raw)
            tensors[EB + ".medians"], tensors[EB + ".quantized_cdf"] = med, cdf
            tensors[EB + ".cdf_length"], tensors[EB + ".offset"] = length, offset
            updated = True
    return updated
