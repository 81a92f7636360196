"""What /root/reference/ntedit.cpp itself holds and a test can pin (VERDICT r5 "next" 2): the candidate tables
(num_tries :172, polish_bases_array / snv_bases_array :176-199, multi_possible_bases :203-348), the opt:: defaults
(:99-133), the literals of the _changes.tsv and _variants.vcf headers (:2165-2211) and of the default output prefix
(:2496-2502).  extract(path) parses the reference's SOURCE TEXT (in the build container only) into canonical sections;
the repository keeps their SHA-256 (tests/golden/reference_tables.json: hashes, not text), so that the GPU tier -- where
/root/reference does not exist -- can still check the product against what the reference holds.

    python tests/tools/reference_tables.py            # prints the sections' hashes
    python tests/tools/reference_tables.py --write    # (re)writes tests/golden/reference_tables.json
"""
import hashlib
import json
import math
import os
import re
import sys

REFERENCE = "/root/reference/ntedit.cpp"
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "reference_tables.json")
LETTERS = "ATCGRYSWKMBDHVN"


def _block(src, start_pat, end_pat=r"\n\};"):
    m = re.search(start_pat, src)
    if not m:
        raise ValueError("reference source: %r not found" % start_pat)
    e = re.compile(end_pat).search(src, m.end())
    if not e:
        raise ValueError("reference source: end of %r not found" % start_pat)
    return src[m.end():e.start()]


def _char_map(body):
    """{ 'A', { 'T', 'C', 'G' } }, ... -> {letter: candidates}"""
    out = {}
    for m in re.finditer(r"\{\s*'(.)'\s*,\s*\{([^}]*)\}\s*\}", body):
        out[m.group(1)] = "".join(re.findall(r"'(.)'", m.group(2)))
    return out


def _literals(text):
    """the C string literals of a stretch of source, in order, unescaped (adjacent literals are NOT joined)"""
    out = []
    text = "\n".join(line for line in text.splitlines() if not line.lstrip().startswith("//"))
    for m in re.finditer(r'"((?:[^"\\]|\\.)*)"', text):
        out.append(m.group(1).encode().decode("unicode_escape"))
    return out


def extract(path=REFERENCE):
    """-> dict of canonical section texts"""
    src = open(path, encoding="utf-8", errors="replace").read()
    sec = {}
    # ---- num_tries
    m = re.search(r"num_tries\s*=\s*\{([^}]*)\}", src)
    sec["num_tries"] = "num_tries " + " ".join(x.strip() for x in m.group(1).split(",")) + "\n"
    # ---- candidate-base tables
    pol = _char_map(_block(src, r"polish_bases_array\s*=\s*\{"))
    snv = _char_map(_block(src, r"snv_bases_array\s*=\s*\{"))
    sec["polish_bases"] = "".join("polish %s %s\n" % (c, pol[c]) for c in LETTERS)
    sec["snv_bases"] = "".join("snv %s %s\n" % (c, snv[c]) for c in LETTERS)
    if set(pol) != set(LETTERS) or set(snv) != set(LETTERS):
        raise ValueError("candidate-base tables: unexpected keys")
    # ---- multi_possible_bases
    body = _block(src, r"multi_possible_bases\s*=\s*\{")
    multi = {}
    for m in re.finditer(r"\{\s*'(.)'\s*,\s*\{([^}]*)\}\s*\}", body):
        multi[m.group(1)] = re.findall(r'"([ACGT]+)"', m.group(2))
    if sorted(multi) != list("ACGT") or any(len(v) != 341 for v in multi.values()):
        raise ValueError("multi_possible_bases: expected 4 x 341 strings")
    sec["multi_possible_bases"] = "".join("multi %s %s\n" % (c, " ".join(multi[c])) for c in "ACGT")
    # ---- opt:: defaults
    ns = _block(src, r"namespace opt \{", r"\} // namespace opt")
    consts = dict(re.findall(r"constexpr\s+\w+\s+(\w+)\s*=\s*([^;]+);", ns))
    vals = {}
    for typ, name, val in re.findall(r"^\s*(?:unsigned|int|float|bool)\s+(\w+)\s*=\s*([^;]+);", ns, re.M) and \
            [(None, n, v) for n, v in re.findall(r"^\s*(?:unsigned|int|float|bool)\s+(\w+)\s*=\s*([^;]+);", ns, re.M)]:
        v = val.strip()
        v = consts.get(v, v)
        vals[name] = v
    want = ["min_contig_len", "max_insertions", "max_deletions", "edit_threshold", "missing_threshold", "edit_ratio",
            "missing_ratio", "use_ratio", "jump", "mode", "snv", "mask", "min_threshold", "max_threshold"]

    def num(v):
        v = {"false": "0", "true": "1"}.get(v, v)
        return "%g" % float(v)
    sec["opt_defaults"] = "".join("%s %s\n" % (n, num(vals[n])) for n in want)
    # ---- _changes.tsv header (ntedit.cpp:2177-2190): its literals, in source order
    tsv = _block(src, r'rfout\.open\(r_filename\);', r'vfout\.open\(v_filename\);')
    sec["tsv_header_literals"] = json.dumps(_literals(tsv)) + "\n"
    # ---- _variants.vcf header (2192-2211): the literals of the vfout << lines
    vcf = _block(src, r'vfout\.open\(v_filename\);', r'#pragma omp parallel')
    lits = []
    for line in vcf.splitlines():
        if "vfout <<" in line:
            lits.extend(_literals(line))
    sec["vcf_header_literals"] = json.dumps(lits) + "\n"
    m = re.search(r'#define\s+PROGRAM\s+"([^"]*)"', src)
    sec["program"] = (m.group(1) if m else "") + "\n"
    # ---- default prefix (2496-2502): its literals
    pre = _block(src, r"if \(opt::outfile_prefix\.empty\(\)\) \{", r"opt::outfile_prefix = outfile_name\.str\(\);")
    sec["prefix_literals"] = json.dumps(_literals(pre)) + "\n"
    sec["prefix_fields"] = " ".join(re.findall(r"<<\s*(?:opt::)?(\w+)", pre)) + "\n"
    return sec


def tables_text(sec):
    """the four table sections in the order the product / oracle dumps have them"""
    return sec["num_tries"] + sec["polish_bases"] + sec["snv_bases"] + sec["multi_possible_bases"]


def split_tables(text):
    """a dump (ora_tables_dump / hostsim_tables_dump / ntedit_hip_device_tables) -> the same four sections"""
    out = {"num_tries": "", "polish_bases": "", "snv_bases": "", "multi_possible_bases": ""}
    key = {"num_tries": "num_tries", "polish": "polish_bases", "snv": "snv_bases", "multi": "multi_possible_bases"}
    for line in text.splitlines(True):
        out[key[line.split(" ", 1)[0]]] += line
    return out


def tsv_header(lits, k, jump, counting):
    """the header line ntedit.cpp:2177-2190 writes, rebuilt from ITS literals (order of appearance in the source:
    0 the four fixed columns, 1 the counting column, 2-4 "Support ", "-mer (out of ", ")", 5 "Support", 6 "Coverage",
    7.. the alternates)"""
    if len(lits) != 13 or lits[5] != "Support" or lits[6] != "Coverage":
        raise ValueError("tsv header: the reference's literals are not the 13 expected")
    evi = lits[6] if counting else lits[5]
    col = lits[1] if counting else lits[2] + str(k) + lits[3] + ("%g" % math.ceil(k / jump)) + lits[4]
    return lits[0] + col + lits[7] + evi + lits[8] + lits[9] + evi + lits[10] + lits[11] + evi + lits[12]


def vcf_header(lits, program, draft, date="00000000"):
    """the lines of ntedit.cpp:2194-2211 from its literals: 0 fileformat, 1 fileDate=, 2 source=, 3 reference=file:,
    4 FORMAT, 5 INFO, 6 #CHROM"""
    if len(lits) != 7:
        raise ValueError("vcf header: the reference's literals are not the 7 expected")
    return "\n".join([lits[0], lits[1] + date, lits[2] + program, lits[3] + draft, lits[4], lits[5], lits[6]]) + "\n"


def hashes(sec):
    return {k: hashlib.sha256(v.encode()).hexdigest() for k, v in sorted(sec.items())}


if __name__ == "__main__":
    s = extract()
    h = hashes(s)
    print(json.dumps(h, indent=1))
    if "--write" in sys.argv:
        os.makedirs(os.path.dirname(GOLDEN), exist_ok=True)
        with open(GOLDEN, "w") as f:
            json.dump({"source": "sha256 of the canonical sections tests/tools/reference_tables.py extracts from the reference's "
                                 "ntedit.cpp (lines 99-133, 172, 176-199, 203-348, 2165-2211, 2496-2502); hashes only, no text",
                       "sha256": h}, f, indent=1)
            f.write("\n")
