"""The sampler tail's top-r truncation (csrc/sampler.hip, ds_sample_tail_kernel: 64-bit keys = order-preserving image of the
log-probability | inverted class index, bitonic sort network in LDS, the mass summed in rank order in a double accumulator and
rounded to float per rank -- what torch's CPU cumsum does for a float tensor --, kept set = a prefix of
the rank order) restated in numpy -- the same key construction, the same compare-exchange network, the same scan -- against the
reference's sort + cumsum (models/dalle_spec.py:158-174, as oracle/diffsound_oracle.py:truncate_top_r restates it).  No GPU: what is
checked is the ALGORITHM the kernel runs (network indices, key order, tie rule, prefix length); tests/test_hip_kernels.py checks
the kernel.  Ties: the reference calls torch.sort without `stable`, so WHICH of several exactly equal log-probabilities survive at
the cut is an implementation detail of torch's sort (it differs between its CPU and CUDA kernels); the kernel's rule since round 1
is "lower class index first" = a stable descending sort, which is what the comparison below uses; without ties at the cut the two
agree bit for bit, and the oracle itself is used for the tie-free cases."""
import numpy as np
import pytest
import torch

from oracle import diffsound_oracle as orc


def _keys(lp):
    """u64 keys of one column (lp float32 [K]): descending key order = descending value, ascending index among ties"""
    bits = (lp + np.float32(0.0)).view(np.uint32)
    u = np.where(bits & np.uint32(0x80000000), ~bits, bits | np.uint32(0x80000000)).astype(np.uint64)
    idx = np.arange(lp.size, dtype=np.uint64)
    return (u << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - idx)


def _bitonic_desc(key):
    """the kernel's network: for k = 2 .. K, j = k / 2 .. 1: pair (i, i | j), descending where (i & k) == 0"""
    key = key.copy()
    K = key.size
    k = 2
    while k <= K:
        j = k >> 1
        while j > 0:
            t = np.arange(K // 2)
            i = ((t & ~(j - 1)) << 1) | (t & (j - 1))
            l = i | j
            a, b = key[i], key[l]
            sw = np.where((i & k) == 0, a < b, a > b)
            key[i], key[l] = np.where(sw, b, a), np.where(sw, a, b)
            j >>= 1
        k <<= 1
    return key


def _truncate_like_the_kernel(lp, r):
    """lp float32 [K] (the K real classes of one column) -> truncated float32 [K]"""
    K = lp.size
    srt = _bitonic_desc(_keys(lp))
    u = (srt >> np.uint64(32)).astype(np.uint32)
    bits = np.where(u & np.uint32(0x80000000), u & np.uint32(0x7FFFFFFF), ~u).astype(np.uint32)
    p = torch.exp(torch.from_numpy(bits.view(np.float32).copy())).numpy()          # torch's fp32 exp, as the oracle
    n_keep, cum = 1, np.float64(0.0)
    for i in range(K):
        cum = cum + np.float64(p[i])                                              # mass of ranks 0 .. i, double accumulator
        if i + 1 < K and np.float32(cum) < np.float32(r):                         # ... rounded to float per rank (torch's CPU cumsum)
            n_keep += 1
        elif i + 1 < K:
            break
    cls = (np.uint64(0xFFFFFFFF) - (srt & np.uint64(0xFFFFFFFF))).astype(np.int64)
    keep = np.zeros(K, dtype=bool)
    keep[cls[:n_keep]] = True
    return np.where(keep, lp, np.float32(-70.0)), srt, cls


@pytest.mark.parametrize("K", [256, 512])
@pytest.mark.parametrize("kind", ["random", "peaked", "tied", "flat"])
def test_sort_and_sequential_mass_equal_the_reference_truncation(K, kind):
    g = torch.Generator().manual_seed(K + len(kind))
    L = 24
    logits = torch.randn(1, K, L, generator=g) * {"random": 2.0, "peaked": 8.0, "tied": 2.0, "flat": 0.01}[kind]
    if kind == "tied":
        logits = (logits * 2).round() / 2                                        # many exact ties, also at the cut
    lp = orc.predict_start_from_logits(logits) if hasattr(orc, "predict_start_from_logits") else None
    if lp is None:
        lp = torch.cat((torch.log_softmax(logits.double(), dim=1).float(), torch.full((1, 1, L), -70.0)), dim=1).clamp(-70.0, 0.0)
    for r in (0.85, 0.5, 0.999):
        if kind == "tied":      # a stable descending sort = the kernel's tie rule (see the module docstring)
            srt_, idx_ = torch.sort(lp, dim=1, descending=True, stable=True)
            inc = torch.exp(srt_).cumsum(dim=1)
            ks = torch.cat((torch.ones_like(inc[:, :1, :], dtype=torch.bool), (inc < r)[:, :-1, :]), dim=1)
            keep = torch.zeros_like(ks).scatter(1, idx_, ks)
            ref = torch.where(keep, lp, torch.full_like(lp, -70.0))[0, :K].numpy()
        else:
            ref = orc.truncate_top_r(lp, r)[0, :K].numpy()                        # [K][L], the K real classes
        for col in range(L):
            mine, srt, cls = _truncate_like_the_kernel(lp[0, :K, col].numpy().copy(), r)
            assert np.all(srt[:-1] > srt[1:])                                      # a strict descending order of distinct keys
            assert sorted(cls.tolist()) == list(range(K))
            assert np.array_equal(mine, ref[:, col]), (kind, K, r, col)
