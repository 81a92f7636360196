"""fixed inputs of the btllib verification kit (deterministic): sequences to hash, a genome, a draft to query"""
import os
import random
import sys

out = sys.argv[1]
rnd = random.Random(20260929)
acgt = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))  # noqa: E731
with open(os.path.join(out, "acgt.txt"), "w") as f:
    f.write("ACATGCATGCA" + acgt(300) + "\n")  # (starts with btllib's own unit-test k-mers)
    f.write(acgt(200).lower() + acgt(100) + "\n")
    s = list(acgt(400))
    for p in (57, 58, 200, 333):
        s[p] = "N"
    f.write("".join(s) + "\n")
    f.write("A" * 150 + "\n")
with open(os.path.join(out, "exotic.txt"), "w") as f:
    s = list(acgt(600))
    for i, ch in enumerate("RYSWKMBDHVUu-*.5nrykm"):
        s[20 + i * 27] = ch
    f.write("".join(s) + "\n")
genome = [acgt(60000), acgt(3000).lower(), acgt(20000)[:10000] + "NNNNNNNN" + acgt(10000)]
with open(os.path.join(out, "genome.fa"), "w") as f:
    for i, g in enumerate(genome):
        f.write(">g%d\n" % i)
        for o in range(0, len(g), 70):
            f.write(g[o:o + 70] + "\n")
with open(os.path.join(out, "draft.txt"), "w") as f:
    d = list(genome[0][1000:9000])
    for p in range(50, len(d), 211):
        d[p] = rnd.choice([c for c in "ACGT" if c != d[p]])
    f.write("".join(d) + "\n")
    f.write(genome[2][9900:10200] + "\n")
