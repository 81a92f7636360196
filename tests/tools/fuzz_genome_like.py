"""TEST INFRASTRUCTURE (runs the CPU oracle as the checker).  GPU only.
Randomised parity hunt on drafts that look like assemblies, at sizes where the partitioned screening runs with its REAL
run sizes (no bin_cap_percent): 72-160 Mbases per case, filters of 128 MiB - 2 GiB (powers of two and not), random shares of
simple-sequence arrays / satellites / dispersed repeats / segmental duplications / novel stretches -- up to half a draft
of simple sequence, where the overflow list of a record chunk runs out and the direct kernel screens the chunk again --
k, hash count, -i / -d / -m drawn at random.  Every contig of every case against the multi-threaded oracle with the same
filter (downloaded from HBM): complete _edited.fa, _changes.tsv, VCF body.
usage: python tests/tools/fuzz_genome_like.py [--minutes M] [--cases N] [--seed S]
Prints one line per case; exit code 1 on any mismatch."""
import argparse
import filecmp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import helpers as H  # noqa: E402


def draw(rng):
    c = dict(bases=float(rng.choice([72e6, 96e6, 128e6, 160e6])), k=int(rng.choice([25, 25, 25, 32, 40])),
             hashes=int(rng.choice([2, 3, 3, 4])),
             filter_bytes=int(rng.choice([1 << 27, 1 << 28, 1 << 29, 1 << 30, 1 << 31, 200_000_008, 300_000_000])),
             contig_len=int(rng.choice([0, 0, 100_000, 1_000_000])))
    heavy = rng.random() < 0.35
    c["fractions"] = dict(simple=float(rng.choice([0.5, 0.3, 0.15]) if heavy else rng.choice([0.0, 0.03, 0.06])),
                          sat=float(rng.choice([0.0, 0.03, 0.10])), dispersed=float(rng.choice([0.0, 0.05, 0.10])),
                          segdup=float(rng.choice([0.0, 0.02, 0.05])), novel=float(rng.choice([0.0, 0.005, 0.02])))
    c["params"] = dict(max_insertions=int(rng.choice([5, 5, 4, 2, 0])), max_deletions=int(rng.choice([5, 5, 9, 3, 0])),
                       mode=int(rng.choice([0, 0, 0, 1])))
    if c["params"]["max_insertions"] == 0:
        c["params"]["max_deletions"] = 0
    c["seed"] = int(rng.integers(1, 1 << 30))
    return c


def run_case(c, tmp):
    import torch
    import ntedit_amd
    from ntedit_amd.synth import SyntheticJob
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_params(ntedit_amd.default_params(**c["params"]))
        job = SyntheticJob(pol, c["bases"], k=c["k"], hash_num=c["hashes"], filter_bytes=c["filter_bytes"], seed=c["seed"],
                           structure="genome", structure_fractions=c["fractions"], contig_len=c["contig_len"])
        nc = len(job.lens)
        names = [b"contig%d" % i for i in range(nc)]
        res = pol.polish_batch(None, job.offsets, job.lens, device_ptr=job.device_ptr, n=job.n_bytes)
        host = job.batch.cpu().numpy()
        fa, tsv, vcf = (os.path.join(tmp, "g" + s) for s in ("_edited.fa", "_changes.tsv", "_variants.vcf"))
        pol.write_tsv_header(tsv)
        open(vcf, "wb").close()
        res.write(host, job.offsets, job.lens, names, fa, tsv, append=True, vcf_path=vcf)
        st = res.stats()
        res.free()
        bits = pol.filter_download(0)
        ofa, otsv, ovcf = (os.path.join(tmp, "o" + s) for s in ("_edited.fa", "_changes.tsv", "_body.vcf"))
        done = H.oracle_polish_flat_mt_files(host, job.offsets, job.lens, names, bits, c["hashes"], c["k"], H.usable_cpus(),
                                             fa_path=ofa, tsv_path=otsv, vcf_path=ovcf, **c["params"])
        same = done == job.n_bases and all(filecmp.cmp(a, b, shallow=False) for a, b in ((fa, ofa), (tsv, otsv), (vcf, ovcf)))
        info = ("%d contigs, screening %s, %d record chunk(s), %d re-screened directly, %d overflow entries, step %.1f ms, %d edits"
                % (nc, "partitioned" if st.screen_binned else "direct", st.screen_launches, st.screen_chunks_direct,
                   st.screen_overflow_records, st.ms_total, st.substitutions + st.insertions + st.deletions))
        del job
        torch.cuda.empty_cache()
        return same, info
    finally:
        pol.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--cases", type=int, default=0)
    ap.add_argument("--seed", type=int, default=606)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.minutes * 60
    n = bad = 0
    while (a.cases and n < a.cases) or (not a.cases and time.time() < t_end):
        c = draw(rng)
        with tempfile.TemporaryDirectory(prefix="fuzzgl_") as tmp:
            ok, info = run_case(c, tmp)
        n += 1
        bad += 0 if ok else 1
        print("%s case %d %s: %s" % ("ok      " if ok else "MISMATCH", n, {k: v for k, v in c.items()}, info), flush=True)
    print("fuzz_genome_like: %d cases, %d mismatches (seed %d)" % (n, bad, a.seed))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
