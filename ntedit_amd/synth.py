"""Deterministic synthetic drafts + Bloom filters, generated on the GPU with
torch (plumbing only: random numbers, masks, repeat_interleave).  Used by
bench.py and by the full-size GPU property tests; layout follows SURVEY.md 8(d):
i.i.d. uniform ACGT truth genome, contigs log-uniform in [50 kbp, 50 Mbp],
draft = truth with substitutions (1e-3) and short indels (5e-5 each, geometric
length mean 1.5 capped at 5), one 1 kbp N-run per 10 Mbp."""
import math

import numpy as np
import torch

ACGT = (65, 67, 71, 84)


def contig_lengths(total_bases, seed, lo=50_000, hi=50_000_000):
    rng = np.random.default_rng(seed)
    out, left = [], int(total_bases)
    hi = max(lo, min(hi, left))
    while left > 0:
        l = int(math.exp(rng.uniform(math.log(lo), math.log(hi))))
        l = min(l, left)
        if left - l < lo:
            l = left
        out.append(l)
        left -= l
    return out


def _geom_len(u, cap=5):
    # geometric, P(len = n) = (2/3)(1/3)^(n-1): mean 1.5; capped
    l = 1 + torch.floor(torch.log(u.clamp_min(1e-12)) / math.log(1.0 / 3.0)).to(torch.int64)
    return l.clamp_(1, cap)


def truth_codes(n, gen, device):
    return torch.randint(0, 4, (n,), generator=gen, device=device, dtype=torch.uint8)


def codes_to_bytes(codes):
    lut = torch.tensor(ACGT, dtype=torch.uint8, device=codes.device)
    return lut[codes.long()]


def mutate_codes(c, gen, p_sub=1e-3, p_ins=5e-5, p_del=5e-5):
    """c: uint8 codes 0..3 on the GPU -> mutated codes"""
    n = c.numel()
    dev = c.device
    r = torch.rand(n, generator=gen, device=dev)
    sub = r < p_sub
    shift = torch.randint(1, 4, (n,), generator=gen, device=dev, dtype=torch.uint8)
    c = torch.where(sub, (c + shift) & 3, c)
    del r, shift, sub
    # deletions
    r = torch.rand(n, generator=gen, device=dev)
    dstart = r < p_del
    dlen = _geom_len(torch.rand(n, generator=gen, device=dev))
    delmask = torch.zeros(n, dtype=torch.bool, device=dev)
    for j in range(5):
        m = dstart & (dlen > j)
        if j:
            delmask[j:] |= m[: n - j]
        else:
            delmask |= m
    del r, dstart, dlen
    # insertions behind a base
    r = torch.rand(n, generator=gen, device=dev)
    istart = r < p_ins
    ilen = _geom_len(torch.rand(n, generator=gen, device=dev))
    counts = 1 + istart.to(torch.int64) * ilen
    counts[delmask] = 0
    del r, istart, ilen, delmask
    out = torch.repeat_interleave(c, counts)
    starts = torch.cumsum(counts, 0) - counts
    first = torch.zeros(out.numel(), dtype=torch.bool, device=dev)
    first[starts[counts > 0]] = True
    ins = ~first
    k = int(ins.sum().item())
    if k:
        out[ins] = torch.randint(0, 4, (k,), generator=gen, device=dev, dtype=torch.uint8)
    return out


class SyntheticJob:
    """Truth genome -> filter in HBM (built with the library's insert kernel) and the
    mutated draft, laid out as one batch (contigs separated by '\\n') in HBM."""

    def __init__(self, polisher, total_bases, k=25, hash_num=3, filter_bytes=1 << 32, seed=20251031,
                 draft_seed=None, device="cuda", build_filter="alloc", n_runs=True, mutate=True,
                 rep_filter_bytes=0, rep_fraction=0.01, contig_len=0):
        """build_filter: "alloc" = allocate a filter in the library and fill it; "insert" = fill the
        filter the polisher already has (e.g. a shared tensor); False = leave the filter alone.
        mutate=False keeps the draft identical to the truth genome (every k-mer is in the filter).
        rep_filter_bytes > 0 also builds a SECONDARY ("repeat", -e) filter holding the k-mers of the first
        rep_fraction of every truth contig.
        contig_len > 0: contigs of exactly that many truth bases (BASELINE.json configs[2]: 2,500 x 100 kbp)
        instead of the log-uniform 50 kbp - 50 Mbp mix."""
        self.total_bases = int(total_bases)
        dev = torch.device(device)
        if contig_len:
            lens = [int(contig_len)] * max(1, int(total_bases) // int(contig_len))
        else:
            lens = contig_lengths(total_bases, seed)
        gen_t = torch.Generator(device=dev)
        gen_t.manual_seed(seed)
        gen_d = torch.Generator(device=dev)
        gen_d.manual_seed((seed + 1) if draft_seed is None else draft_seed)
        if build_filter == "alloc":
            polisher.filter_alloc(filter_bytes, hash_num, k)
        if rep_filter_bytes:
            polisher.filter_alloc(rep_filter_bytes, hash_num, k, slot=1)
        parts, offs, dlens = [], [], []
        pos = 0
        nl = torch.tensor([10], dtype=torch.uint8, device=dev)
        for L in lens:
            t = truth_codes(L, gen_t, dev)
            if build_filter:
                tb = codes_to_bytes(t)
                torch.cuda.synchronize(dev)
                polisher.filter_insert(None, device_ptr=tb.data_ptr(), n=tb.numel())
                if rep_filter_bytes:
                    polisher.filter_insert(None, slot=1, device_ptr=tb.data_ptr(), n=max(k, int(tb.numel() * rep_fraction)))
                del tb
            d = codes_to_bytes(mutate_codes(t, gen_d) if mutate else t)
            if n_runs:
                for p in range(5_000_000, d.numel() - 1000, 10_000_000):
                    d[p:p + 1000] = 78  # 'N'
            offs.append(pos)
            dlens.append(d.numel())
            parts.append(d)
            parts.append(nl)
            pos += d.numel() + 1
            del t
        self.batch = torch.cat(parts)
        del parts
        self.offsets = np.array(offs, dtype=np.uint64)
        self.lens = np.array(dlens, dtype=np.uint32)
        self.n_bases = int(self.lens.astype(np.int64).sum())
        torch.cuda.synchronize(dev)

    @property
    def device_ptr(self):
        return self.batch.data_ptr()

    @property
    def n_bytes(self):
        return self.batch.numel()


def counting_filter_from_plain(polisher, k, hash_num, device="cuda", seed=7):
    """Turns the PLAIN primary filter the polisher holds (n bytes = 8n bit slots) into a counting filter of 8n
    8-bit counters in HBM and makes it the polisher's primary filter: a k-mer's counting slots hv % (8n) are its bit
    slots, so every truth k-mer gets non-zero counters.  Counter values are 1..4, a fixed function of the slot
    index, so that -p 2 / -q cut into the present k-mers.  (Synthetic content for throughput / parity runs: the
    conservative-update insertion of btllib's KmerCountingBloomFilter8 is sequential and stays on the host,
    oracle/mkbf.c.)  Returns the counter tensor; the caller keeps it alive while the polisher uses it."""
    dev = torch.device(device)
    bits = torch.from_numpy(polisher.filter_download(0)).to(dev)
    n = bits.numel()
    counters = torch.empty(n * 8, dtype=torch.uint8, device=dev)
    view = counters.view(n, 8)
    step = 1 << 24
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        b = bits[lo:hi].to(torch.int32)
        idx = torch.arange(lo * 8, hi * 8, device=dev, dtype=torch.int64).view(-1, 8)
        val = (1 + ((idx * 2654435761 + seed) >> 13) % 4).to(torch.uint8)
        for j in range(8):
            view[lo:hi, j] = torch.where(((b >> j) & 1) != 0, val[:, j], torch.zeros_like(val[:, j]))
        del b, idx, val
    del bits
    torch.cuda.synchronize(dev)
    polisher.set_filter_device(counters.data_ptr(), counters.numel(), hash_num, k, counting=True)
    return counters
