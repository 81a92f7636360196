"""What the reference's own source holds, pinned (VERDICT r5 "next" 2).

In the build container (/root/reference present) tests/tools/reference_tables.py parses ntedit.cpp -- the 4 x 341
multi_possible_bases strings, num_tries, both candidate-base tables, the opt:: defaults, the literals of the TSV / VCF headers
and of the default prefix -- and these tests compare that TEXT with the oracle's restatement, with the product's device
functions (through the host build, tests/hostsim) and host-side writers, and with the committed SHA-256 of each section
(tests/golden/reference_tables.json).  Where the reference is absent (the GPU box) the same product outputs are checked
against the committed hashes: hashes travel, the reference's text does not."""
import ctypes
import hashlib
import json
import os
import re
import sys

import pytest

import helpers as H

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
import reference_tables as R  # noqa: E402

HAVE_REF = os.path.exists(R.REFERENCE)
GOLD = json.load(open(R.GOLDEN))["sha256"]
TABLE_SECTIONS = ("num_tries", "polish_bases", "snv_bases", "multi_possible_bases")


def _sha(text):
    return hashlib.sha256(text.encode()).hexdigest()


def _oracle_tables():
    lib = H.oracle_lib()
    lib.ora_tables_dump.restype = ctypes.c_long
    buf = ctypes.create_string_buffer(1 << 16)
    n = lib.ora_tables_dump(buf, ctypes.c_size_t(len(buf)))
    assert n > 0
    return buf.raw[:n].decode()


def _hostsim_tables():
    hs = ctypes.CDLL(H.build_hostsim())
    hs.hostsim_tables_dump.restype = ctypes.c_long
    buf = ctypes.create_string_buffer(1 << 16)
    n = hs.hostsim_tables_dump(buf, ctypes.c_size_t(len(buf)))
    assert n > 0
    return buf.raw[:n].decode()


def _product_params_text():
    import ntedit_amd
    p = ntedit_amd.default_params()
    names = ["min_contig_len", "max_insertions", "max_deletions", "edit_threshold", "missing_threshold", "edit_ratio",
             "missing_ratio", "use_ratio", "jump", "mode", "snv", "mask", "min_threshold", "max_threshold"]
    return "".join("%s %g\n" % (n, float(getattr(p, n))) for n in names)


def _product_headers(tmp_path):
    from ntedit_amd import _lib
    lib = _lib.load()
    out = {}
    for counting in (0, 1):
        path = str(tmp_path / ("h%d.tsv" % counting))
        assert lib.ntedit_hip_write_tsv_header(path.encode(), 25, 3, counting) == 0
        out["tsv%d" % counting] = open(path).read()
    path = str(tmp_path / "h.vcf")
    assert lib.ntedit_hip_write_vcf_header(path.encode(), b"some/draft.fa") == 0
    out["vcf"] = re.sub(r"(##fileDate=)\d{8}", r"\g<1>00000000", open(path).read())
    return out


def _main_cpp_prefix():
    """the default-prefix expression of host/main.cpp as {literals, fields}"""
    src = open(os.path.join(H.ROOT, "ntedit_amd", "host", "main.cpp")).read()
    m = re.search(r"if \(prefix\.empty\(\)\) \{(.*?)prefix = o\.str\(\);", src, re.S)
    body = "\n".join(l for l in m.group(1).splitlines() if not l.lstrip().startswith("//"))
    lits = re.findall(r'"((?:[^"\\]|\\.)*)"', body)
    fields = re.findall(r"<<\s*(?:base_name\()?(?:p\.)?(\w+)", body)
    return lits, [f for f in fields if f != "o"]


# ---------------------------------------------------------------- against the reference's text (build container)
@pytest.mark.skipif(not HAVE_REF, reason="/root/reference is not on this machine")
def test_committed_hashes_are_the_reference(oracle_build):
    sec = R.extract()
    assert R.hashes(sec) == GOLD, "tests/golden/reference_tables.json is stale: python tests/tools/reference_tables.py --write"


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference is not on this machine")
def test_candidate_tables_equal_the_reference_text(oracle_build):
    """all 4 x 341 insertion strings in the reference's order, num_tries, polish / snv candidate bases: reference source text ==
    the oracle's generator == the product's device functions (host build)"""
    sec = R.extract()
    want = R.tables_text(sec)
    ora, prod = _oracle_tables(), _hostsim_tables()
    assert ora == want
    assert prod == want
    assert want.count(" ") > 4 * 341


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference is not on this machine")
def test_defaults_and_headers_equal_the_reference_text(tmp_path, oracle_build):
    sec = R.extract()
    assert _product_params_text() == sec["opt_defaults"]
    lib = H.oracle_lib()
    op = H.OraParams()
    lib.ora_params_default(ctypes.byref(op))
    for line in sec["opt_defaults"].splitlines():
        name, val = line.split()
        assert float(getattr(op, name)) == float(val), name
    tl = json.loads(sec["tsv_header_literals"])
    vl = json.loads(sec["vcf_header_literals"])
    got = _product_headers(tmp_path)
    assert got["tsv0"] == R.tsv_header(tl, 25, 3, False)
    assert got["tsv1"] == R.tsv_header(tl, 25, 3, True)
    assert got["vcf"] == R.vcf_header(vl, sec["program"].strip(), "some/draft.fa")
    lits, fields = _main_cpp_prefix()
    assert lits == json.loads(sec["prefix_literals"])
    ref_fields = sec["prefix_fields"].split()
    # draft_basename k min_contig_len bloom_basename max_insertions max_deletions mode
    assert [f.replace("draft", "draft_basename").replace("bf", "bloom_basename") if f in ("draft", "bf") else f
            for f in fields] == ref_fields
    assert lib is not None


# ---------------------------------------------------------------- against the committed hashes (any machine)
def test_tables_match_the_committed_hashes(oracle_build):
    for name, text in (("oracle", _oracle_tables()), ("product", _hostsim_tables())):
        parts = R.split_tables(text)
        for s in TABLE_SECTIONS:
            assert _sha(parts[s]) == GOLD[s], (name, s)


def test_defaults_and_headers_match_the_committed_hashes(tmp_path):
    assert _sha(_product_params_text()) == GOLD["opt_defaults"]
    got = _product_headers(tmp_path)
    # the header lines hold the reference's literals in the reference's order: hash of the literal list rebuilt from the output
    m = re.fullmatch(r"(ID\tbpPosition\+1\tOriginalBase\tNewBase\t)(Support )25(-mer \(out of )9(\))(\tAlt\.Base1\tAlt\.)(Support)(1\t)"
                     r"(Alt\.Base2\tAlt\.)Support(2\t)(Alt\.Base3\tAlt\.)Support(3\n)", got["tsv0"])
    assert m
    c = re.fullmatch(r"ID\tbpPosition\+1\tOriginalBase\tNewBase\t(Coverage \(max 255\))\tAlt\.Base1\tAlt\.(Coverage)1\t.*", got["tsv1"], re.S)
    assert c
    g = m.groups()
    lits = [g[0], c.group(1), g[1], g[2], g[3], g[5], c.group(2), g[4], g[6], g[7], g[8], g[9], g[10]]
    assert _sha(json.dumps(lits) + "\n") == GOLD["tsv_header_literals"]
    lines = got["vcf"].splitlines()
    assert len(lines) == 7
    prog = lines[2][len("##source="):]
    vl = [lines[0], "##fileDate=", "##source=", "##reference=file:", lines[4], lines[5], lines[6]]
    assert lines[1] == "##fileDate=00000000" and lines[3] == "##reference=file:some/draft.fa"
    assert _sha(json.dumps(vl) + "\n") == GOLD["vcf_header_literals"]
    assert _sha(prog + "\n") == GOLD["program"]
    lits, fields = _main_cpp_prefix()
    assert _sha(json.dumps(lits) + "\n") == GOLD["prefix_literals"]


@pytest.mark.gpu
def test_device_tables_match_the_committed_hashes():
    """the tables as the GPU's own code produces them (ntedit_hip_device_tables: one thread runs MachineT::candidate_bases /
    insertion_candidate on the device)"""
    import ntedit_amd
    pol = ntedit_amd.Polisher(0)
    try:
        text = pol.device_tables()
    finally:
        pol.close()
    parts = R.split_tables(text)
    for s in TABLE_SECTIONS:
        assert _sha(parts[s]) == GOLD[s], s
    if HAVE_REF:
        assert text == R.tables_text(R.extract())
