"""python -m ntedit_amd.merge -o OUT_PREFIX SHARD_PREFIX [SHARD_PREFIX ...]

Gathers the outputs of `ntedit --shard I/N -b SHARD_PREFIX` processes (one per GPU) into OUT_PREFIX_edited.fa /
_changes.tsv / _variants.vcf in input order (= the reference at -t 1).  Every shard wrote SHARD_PREFIX.index.tsv: the
ordinal of each of its contigs in the draft and the bytes it appended to the three files; the gather copies byte
ranges in ordinal order -- rows are never matched by contig name (names need not be unique)."""
import argparse
import sys

from .dist import _copy_range


# what ntedit_hip_write_tsv_header / ntedit_hip_write_vcf_header put in front of the records (host/render.cpp):
# counted in lines, never sniffed by prefix -- a contig may be called "#chr1"
TSV_HEADER_LINES = 1
VCF_HEADER_LINES = 7


def _header_bytes(path, n_lines):
    n = 0
    with open(path, "rb") as f:
        for _ in range(n_lines):
            n += len(f.readline())
    return n


def merge_cli_shards(out_prefix, shard_prefixes):
    entries = []  # (ordinal, shard, fa off, tsv off, vcf off, sizes)
    for s, pre in enumerate(shard_prefixes):
        off = [0,
               _header_bytes(pre + "_changes.tsv", TSV_HEADER_LINES),
               _header_bytes(pre + "_variants.vcf", VCF_HEADER_LINES)]
        with open(pre + ".index.tsv", "rb") as f:
            for line in f:
                if line.startswith(b"#"):
                    continue
                o, nf, nt, nv = (int(x) for x in line.split())
                entries.append((o, s, tuple(off), (nf, nt, nv)))
                off[0] += nf
                off[1] += nt
                off[2] += nv
    entries.sort()
    if [e[0] for e in entries] != list(range(len(entries))):
        raise SystemExit("ntedit_amd.merge: the shard indexes do not cover the draft's contigs exactly once")
    suffixes = ("_edited.fa", "_changes.tsv", "_variants.vcf")
    ins = [[open(pre + suf, "rb") for suf in suffixes] for pre in shard_prefixes]
    outs = [open(out_prefix + suf, "wb") for suf in suffixes]
    try:
        first = shard_prefixes[0]
        for j, hdr in ((1, TSV_HEADER_LINES), (2, VCF_HEADER_LINES)):
            n = _header_bytes(first + suffixes[j], hdr)
            _copy_range(ins[0][j], outs[j], n)
        for _, s, off, sz in entries:
            for j in range(3):
                ins[s][j].seek(off[j])
                _copy_range(ins[s][j], outs[j], sz[j])
    finally:
        for f in outs:
            f.close()
        for fs in ins:
            for f in fs:
                f.close()
    return len(entries)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m ntedit_amd.merge", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-o", dest="out", required=True)
    ap.add_argument("shards", nargs="+")
    a = ap.parse_args(argv)
    n = merge_cli_shards(a.out, a.shards)
    sys.stderr.write("ntedit_amd.merge: %d contigs from %d shard(s)\n" % (n, len(a.shards)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
