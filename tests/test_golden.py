"""Committed golden vectors (tests/golden/cases, made by tests/golden/make_golden.py):
the oracle and the host build of the product's control logic must reproduce them
byte for byte; the GPU path is checked against the same files in test_gpu_parity."""
import filecmp
import os
import shlex

import pytest

import helpers as H

CASES = sorted(os.listdir(os.path.join(H.GOLDEN, "cases")))


def params_from_file(path):
    toks = shlex.split(open(path).read())
    kw = {}
    key = {"-z": ("min_contig_len", int), "-i": ("max_insertions", int), "-d": ("max_deletions", int),
           "-j": ("jump", int), "-m": ("mode", int), "-a": ("mask", int), "-p": ("min_threshold", int),
           "-q": ("max_threshold", int), "-s": ("snv", int), "-x": ("missing_threshold", float),
           "-y": ("edit_threshold", float),
           "-X": ("missing_ratio", float), "-Y": ("edit_ratio", float)}
    for flag, val in zip(toks[0::2], toks[1::2]):
        name, conv = key[flag]
        kw[name] = conv(val)
        if flag in ("-X", "-Y"):
            kw["use_ratio"] = 1
    return H.default_params(**kw)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(tmp_path, name, oracle_build):
    d = os.path.join(H.GOLDEN, "cases", name)
    hp = params_from_file(os.path.join(d, "params.txt"))
    rep = os.path.join(d, "secondary.bf")
    H.run_oracle(os.path.join(d, "draft.fa"), os.path.join(d, "filter.bf"), hp, str(tmp_path / "o"),
                 rep if os.path.exists(rep) else None)
    assert filecmp.cmp(os.path.join(d, "expected_changes.tsv"), str(tmp_path / "o_changes.tsv"), shallow=False)
    assert filecmp.cmp(os.path.join(d, "expected_edited.fa"), str(tmp_path / "o_edited.fa"), shallow=False)
    assert open(os.path.join(d, "expected_variants.vcf.body")).read().splitlines() == \
        H.vcf_body(str(tmp_path / "o_variants.vcf"))


@pytest.mark.parametrize("name", CASES)
def test_hostsim_reproduces_golden(tmp_path, name, oracle_build):
    d = os.path.join(H.GOLDEN, "cases", name)
    hp = params_from_file(os.path.join(d, "params.txt"))
    rep = os.path.join(d, "secondary.bf")
    rc, _, _ = H.run_hostsim(H.read_fasta(os.path.join(d, "draft.fa")), H.load_bf(os.path.join(d, "filter.bf")), hp,
                             str(tmp_path / "h"), H.load_bf(rep) if os.path.exists(rep) else None)
    assert rc == 0
    assert filecmp.cmp(os.path.join(d, "expected_changes.tsv"), str(tmp_path / "h_changes.tsv"), shallow=False)
    assert filecmp.cmp(os.path.join(d, "expected_edited.fa"), str(tmp_path / "h_edited.fa"), shallow=False)
    assert open(os.path.join(d, "expected_variants.vcf.body")).read().splitlines() == \
        H.vcf_body(str(tmp_path / "h_variants.vcf"))
