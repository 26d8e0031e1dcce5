"""CPU restatement of the entropy-table builders (TEST INFRASTRUCTURE: only tests/ may import this; the product's
builders are demo-learned-point-cloud-compression_amd/tables.py).

What `model.update()` does in the reference (sender/encoder/codec_pipeline.py:69): CompressAI 1.2.4's
EntropyBottleneck.update / GaussianConditional.update over pmf_to_quantized_cdf.  CompressAI is not in the reference
tree and not installed here: restated from its published sources ([RECALL]) — parity unpinned, like the
rest of oracle/.  Written row by row and element by element, on purpose unlike the product's vectorised form, so that
the two check each other (tests/test_cpu.py).
"""
import math

import numpy as np

f32 = np.float32


def pmf_to_quantized_cdf(pmf, precision=16):
    """ops.cpp pmf_to_quantized_cdf, statement by statement"""
    cdf = [0]
    for p in pmf:
        # std::round(float) for p >= 0: half away from zero.  The product is a float32; the + 0.5 must NOT be (the float just
        # below 0.5 plus 0.5 rounds to 1.0 in float32 where std::round gives 0): it is exact in double
        cdf.append(int(math.floor(float(f32(f32(p) * f32(1 << precision))) + 0.5)))
    total = sum(cdf)
    cdf = [((1 << precision) * c) // total for c in cdf]
    for i in range(1, len(cdf)):               # std::partial_sum
        cdf[i] += cdf[i - 1]
    cdf[-1] = 1 << precision
    for i in range(len(cdf) - 1):
        if cdf[i] == cdf[i + 1]:
            best_freq, best = None, -1
            for j in range(len(cdf) - 1):
                freq = cdf[j + 1] - cdf[j]
                if freq > 1 and (best_freq is None or freq < best_freq):
                    best_freq, best = freq, j
            assert best != -1
            if best < i:
                for j in range(best + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best + 1):
                    cdf[j] += 1
    return np.asarray(cdf, dtype=np.int32)


def gaussian_tables(scale_table, tail_mass=1e-9):
    from scipy.special import erfc
    from scipy.stats import norm
    mult = f32(-norm.ppf(tail_mass / 2))
    rows, lens, offs = [], [], []
    for s in np.asarray(scale_table, dtype=f32):
        center = int(np.ceil(f32(s * mult)))
        length = 2 * center + 1
        pmf = []
        for i in range(length):
            a = f32(abs(i - center))
            up = f32(0.5) * f32(erfc(f32(f32(-(2 ** -0.5)) * f32(f32(f32(0.5) - a) / s))))
            lo = f32(0.5) * f32(erfc(f32(f32(-(2 ** -0.5)) * f32(f32(f32(-0.5) - a) / s))))
            pmf.append(f32(up - lo))
        a0 = f32(center)
        tail = f32(2) * (f32(0.5) * f32(erfc(f32(f32(-(2 ** -0.5)) * f32(f32(f32(-0.5) - a0) / s)))))
        rows.append(pmf_to_quantized_cdf(pmf + [tail]))
        lens.append(length + 2)
        offs.append(-center)
    width = max(len(r) for r in rows)
    cdfs = np.zeros((len(rows), width), dtype=np.int32)
    for i, r in enumerate(rows):
        cdfs[i, :len(r)] = r
    return cdfs, np.asarray(lens, np.int32), np.asarray(offs, np.int32)


def _softplus(x):
    x = f32(x)
    return x if x > f32(20) else f32(np.log1p(np.exp(x)))


def _sigmoid(x):
    return f32(f32(1) / (f32(1) + np.exp(-f32(x))))


def _logits(x, c, mats, biases, factors):
    """_logits_cumulative of channel c at the scalar x"""
    v = np.asarray([f32(x)], dtype=f32)
    for i, (m, b) in enumerate(zip(mats, biases)):
        w = np.vectorize(_softplus, otypes=[f32])(m[c])                     # [f(i+1), f(i)]
        v = (np.matmul(w, v.reshape(-1, 1)).astype(f32) + b[c]).astype(f32).reshape(-1)
        if i < len(factors):
            v = (v + np.tanh(factors[i][c].reshape(-1)) * np.tanh(v)).astype(f32)
    return f32(v[0])


def bottleneck_tables(mats, biases, factors, quantiles):
    q = np.asarray(quantiles, dtype=f32)
    ch = q.shape[0]
    med = [f32(q[c, 0, 1]) for c in range(ch)]
    minima = [max(int(np.ceil(f32(med[c] - q[c, 0, 0]))), 0) for c in range(ch)]
    maxima = [max(int(np.ceil(f32(q[c, 0, 2] - med[c]))), 0) for c in range(ch)]
    lengths = [minima[c] + maxima[c] + 1 for c in range(ch)]
    max_len = max(lengths)
    rows = []
    for c in range(ch):
        start = f32(med[c] - f32(minima[c]))
        pmf = []
        for i in range(lengths[c]):
            x = f32(start + f32(i))
            lo, up = _logits(f32(x - f32(0.5)), c, mats, biases, factors), _logits(f32(x + f32(0.5)), c, mats, biases, factors)
            sign = -np.sign(f32(lo + up))
            pmf.append(f32(abs(f32(_sigmoid(f32(sign * up)) - _sigmoid(f32(sign * lo))))))
        first, last = f32(start), f32(start + f32(max_len - 1))
        tail = f32(_sigmoid(_logits(f32(first - f32(0.5)), c, mats, biases, factors)) +
                   _sigmoid(-_logits(f32(last + f32(0.5)), c, mats, biases, factors)))
        rows.append(pmf_to_quantized_cdf(pmf + [tail]))
    cdfs = np.zeros((ch, max_len + 2), dtype=np.int32)
    for c, r in enumerate(rows):
        cdfs[c, :len(r)] = r
    return (np.asarray(med, f32), cdfs, np.asarray([n + 2 for n in lengths], np.int32),
            np.asarray([-m for m in minima], np.int32))
