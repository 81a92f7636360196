"""Deterministic synthetic drafts + Bloom filters, generated on the GPU with
torch (plumbing only: random numbers, masks, repeat_interleave).  Used by
bench.py and by the full-size GPU property tests; layout follows SURVEY.md 8(d):
i.i.d. uniform ACGT truth genome, contigs log-uniform in [50 kbp, 50 Mbp],
draft = truth with substitutions (1e-3) and short indels (5e-5 each, geometric
length mean 1.5 capped at 5), one 1 kbp N-run per 10 Mbp.

structure="genome" (round 6) lays what assemblies are made of over the i.i.d. truth genome before the draft is derived
from it: simple-sequence arrays, satellite arrays of 171-bp monomers, dispersed repeat families, segmental
duplications, and "novel" stretches the draft has and the filter does not (regions the reads never covered) --
`GenomeStructure` below."""
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


def _loguniform(rng, lo, hi):
    return int(math.exp(rng.uniform(math.log(lo), math.log(hi))))


def _diverge(c, frac, gen):
    """substitutes a fraction `frac` (scalar, or one value per row of a 2-D tensor) of the codes of c"""
    r = torch.rand(c.shape, generator=gen, device=c.device)
    sub = r < frac
    shift = torch.randint(1, 4, c.shape, generator=gen, device=c.device, dtype=torch.uint8)
    return torch.where(sub, (c + shift) & 3, c)


def _revcomp(c):
    return 3 - torch.flip(c, dims=(-1,))


class GenomeStructure:
    """What a genome has and an i.i.d. sequence has not, as fractions of every contig's bases:

      simple   ~3 %  simple-sequence arrays: a unit of 2-6 bases repeated over 10 kbp - 5 Mbp, 1-2 % of the bases substituted
      sat      ~3 %  satellite arrays: one of four 171-bp monomers repeated over 10 kbp - 5 Mbp, 2-5 % substituted
      dispersed ~5 % copies of a 300-bp and of a 6-kbp family (half the bases each), 10-15 % substituted per copy, either strand
      segdup   ~2 %  copies of 10-200 kbp of the same contig, either strand, 1-2 % substituted
      novel    ~0.5 % stretches of 2-50 kbp that stay in the draft and are kept OUT of the filter

    Arrays longer than a quarter of their contig are shortened to that; a feature that would overshoot what is left of
    its class's share is placed with the matching probability, so small contigs get their share on average.  Everything
    is a function of (seed, contig index): the same truth genome on every rank and in every test."""

    def __init__(self, seed, device, fractions=None):
        self.frac = dict(simple=0.03, sat=0.03, dispersed=0.05, segdup=0.02, novel=0.005)
        if fractions:
            self.frac.update(fractions)
        self.seed = int(seed)
        self.device = device
        g = torch.Generator(device=device)
        g.manual_seed(self.seed ^ 0x5A7E111E)
        self.monomers = [truth_codes(171, g, device) for _ in range(4)]
        self.short_family = truth_codes(300, g, device)
        self.long_family = truth_codes(6000, g, device)
        self.bases = dict(simple=0, sat=0, dispersed=0, segdup=0, novel=0)
        self.features = []  # (contig, start, length, class) of the arrays / duplications / novel stretches (diagnosis)

    def _spans(self, rng, L, share, lo, hi):
        """feature lengths of one class for a contig of L bases"""
        out, left = [], share * L
        cap = max(lo, L // 4)
        while left > 0:
            n = min(_loguniform(rng, lo, hi), cap, L)
            if n > left and rng.random() >= left / n:
                break
            out.append(n)
            left -= n
        return out

    def apply(self, t, index, gen):
        """t: codes of truth contig `index` (modified in place) -> boolean mask of its novel stretches (or None)"""
        L = t.numel()
        dev = t.device
        rng = np.random.default_rng((self.seed, int(index)))
        # dispersed repeats: all copies of a family in one scatter
        for fam in (self.short_family, self.long_family):
            n = fam.numel()
            if L < 4 * n:
                continue
            copies = int(self.frac["dispersed"] * 0.5 * L / n + rng.random())
            if copies == 0:
                continue
            starts = torch.from_numpy(rng.integers(0, L - n, size=copies)).to(dev)
            div = torch.from_numpy(rng.uniform(0.10, 0.15, size=copies).astype(np.float32)).to(dev)
            rows = _diverge(fam.unsqueeze(0).expand(copies, n).contiguous(), div.unsqueeze(1), gen)
            flip = torch.from_numpy(rng.random(copies) < 0.5).to(dev)
            rows = torch.where(flip.unsqueeze(1), _revcomp(rows), rows)
            idx = (starts.unsqueeze(1) + torch.arange(n, device=dev).unsqueeze(0)).reshape(-1)
            t[idx] = rows.reshape(-1)
            self.bases["dispersed"] += copies * n
            del rows, idx
        for n in self._spans(rng, L, self.frac["simple"], 10_000, 5_000_000):
            period = int(rng.integers(2, 7))
            unit = rng.integers(0, 4, size=period)
            if (unit == unit[0]).all():
                unit[-1] = (unit[0] + 1) & 3
            u = torch.from_numpy(unit.astype(np.uint8)).to(dev)
            arr = u.repeat((n + period - 1) // period)[:n]
            at = int(rng.integers(0, L - n + 1))
            t[at:at + n] = _diverge(arr, float(rng.uniform(0.01, 0.02)), gen)
            self.bases["simple"] += n
            self.features.append((index, at, n, "simple%d" % period))
        for n in self._spans(rng, L, self.frac["sat"], 10_000, 5_000_000):
            m = self.monomers[int(rng.integers(0, len(self.monomers)))]
            arr = m.repeat((n + 170) // 171)[:n]
            at = int(rng.integers(0, L - n + 1))
            t[at:at + n] = _diverge(arr, float(rng.uniform(0.02, 0.05)), gen)
            self.bases["sat"] += n
            self.features.append((index, at, n, "sat"))
        for n in self._spans(rng, L, self.frac["segdup"], 10_000, 200_000):
            src = int(rng.integers(0, L - n + 1))
            dst = int(rng.integers(0, L - n + 1))
            piece = _diverge(t[src:src + n].clone(), float(rng.uniform(0.01, 0.02)), gen)
            t[dst:dst + n] = _revcomp(piece) if rng.random() < 0.5 else piece
            self.bases["segdup"] += n
            self.features.append((index, dst, n, "segdup"))
        spans = self._spans(rng, L, self.frac["novel"], 2_000, 50_000)
        if not spans:
            return None
        novel = torch.zeros(L, dtype=torch.bool, device=dev)
        for n in spans:
            at = int(rng.integers(0, L - n + 1))
            novel[at:at + n] = True
            self.bases["novel"] += n
            self.features.append((index, at, n, "novel"))
        return novel


class SyntheticJob:
    """Truth genome -> filter in HBM (built with the library's insert kernel) and the
    mutated draft, laid out as one batch (contigs separated by '\\n') in HBM."""

    def __init__(self, polisher, total_bases, k=25, hash_num=3, filter_bytes=1 << 32, seed=20251031,
                 draft_seed=None, device="cuda", build_filter="alloc", n_runs=True, mutate=True,
                 rep_filter_bytes=0, rep_fraction=0.01, contig_len=0, structure="iid", structure_fractions=None):
        """build_filter: "alloc" = allocate a filter in the library and fill it; "insert" = fill the
        filter the polisher already has (e.g. a shared tensor); False = leave the filter alone.
        mutate=False keeps the draft identical to the truth genome (every k-mer is in the filter).
        rep_filter_bytes > 0 also builds a SECONDARY ("repeat", -e) filter holding the k-mers of the first
        rep_fraction of every truth contig.
        contig_len > 0: contigs of exactly that many truth bases (BASELINE.json configs[2]: 2,500 x 100 kbp)
        instead of the log-uniform 50 kbp - 50 Mbp mix.
        structure: "iid" = SURVEY 8(d)'s uniform truth genome; "genome" = the same with `GenomeStructure` laid over it
        (structure_fractions overrides single shares, e.g. {"novel": 0.0})."""
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
        if structure not in ("iid", "genome"):
            raise ValueError("structure must be 'iid' or 'genome'")
        self.structure = GenomeStructure(seed, dev, structure_fractions) if structure == "genome" else None
        parts, offs, dlens = [], [], []
        pos = 0
        nl = torch.tensor([10], dtype=torch.uint8, device=dev)
        for ci, L in enumerate(lens):
            t = truth_codes(L, gen_t, dev)
            novel = self.structure.apply(t, ci, gen_t) if self.structure is not None else None
            if build_filter:
                tb = codes_to_bytes(t)
                if novel is not None:
                    tb[novel] = 78  # the filter never sees the k-mers of a novel stretch
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
            del t, novel
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
