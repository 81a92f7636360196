"""Reconstruct the reference's expected polished E. coli genome.

The reference's demo (demo/runme.sh:8-10) diffs ntEdit's output against
demo/ecoli_ntedit_k25_edited.fa and demo/ecoli_ntedit_k25_changes.tsv.  The
FASTA is a missing large blob in the reference tree, but it is fully determined
by the demo draft + the committed changes.tsv under the row conventions of
writeEditsToFile (ntedit.cpp:957-973, 992-1053, 1201-1203):

  SUB  ID  pos+1  draft  new   support [alt...]
  INS  ID  pos    char   +BASES support    pos = 0-based index of the base after the insertion
  DEL  ID  pos    char   -BASES support    pos = 0-based start of the deleted run

Both inputs are data fixtures copied from the reference's demo directory into
tests/golden/demo/.  Usage: recon_demo.py draft.fa.gz changes.tsv out.fa
"""
import gzip
import sys


def read_fasta(path):
    op = gzip.open if path.endswith(".gz") else open
    name, chunks = None, []
    with op(path, "rt") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(chunks)
                name, chunks = line[1:], []
            else:
                chunks.append(line)
    if name is not None:
        yield name, "".join(chunks)


def reconstruct(draft, rows):
    out = []
    cur = 0
    bad = 0
    for r in rows:
        pos, orig, new = int(r[1]), r[2], r[3]
        if new.startswith("+"):
            out.append(draft[cur:pos])
            out.append(new[1:])
            cur = pos
        elif new.startswith("-"):
            n = len(new) - 1
            if draft[pos:pos + n].upper() != new[1:].upper():
                bad += 1
            out.append(draft[cur:pos])
            cur = pos + n
        else:
            p = pos - 1
            if draft[p].upper() != orig.upper():
                bad += 1
            out.append(draft[cur:p])
            out.append(new)
            cur = p + 1
    out.append(draft[cur:])
    return "".join(out), bad


def main():
    draft_path, tsv_path, out_path = sys.argv[1:4]
    (name, draft), = list(read_fasta(draft_path))
    rows = [l.rstrip("\n").split("\t") for l in open(tsv_path)][1:]
    seq, bad = reconstruct(draft, rows)
    with open(out_path, "w") as f:
        f.write(">%s\n%s\n" % (name, seq))
    print("draft %d bp -> edited %d bp, %d rows, %d convention mismatches" % (len(draft), len(seq), len(rows), bad))


if __name__ == "__main__":
    main()
