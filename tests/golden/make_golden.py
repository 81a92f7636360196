"""Regenerates the small golden cases under tests/golden/cases/.

The reference (ntedit.cpp) cannot be built in this environment (btllib / Boost are
absent and stand-in headers are not allowed), so these vectors come from the CPU
oracle AFTER it was pinned against the reference's own demo fixture
(tests/test_oracle_demo.py).  They freeze the oracle's behaviour so that an
accidental change of the restatement shows up as a diff; the device path is
compared against the same files on the GPU.

usage: python tests/golden/make_golden.py     (rewrites cases/*/expected_*)
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402

CASES = {
    # name: (make_case kwargs, parameter kwargs)
    "default_k25": (dict(n=12000, contigs=2, p_sub=3e-3, p_ins=6e-4, p_del=6e-4, bfbytes=1 << 14), dict()),
    "mode1_mask_k31_nonpow2": (dict(n=12000, contigs=2, k=31, hashes=4, bfbytes=20011 * 8 // 8 * 8, flavor="N lower"),
                               dict(mode=1, mask=1, max_deletions=8)),
    "secondary_ratio": (dict(n=12000, contigs=2, flavor="sec iupac", bfbytes=1 << 14), dict(use_ratio=1, jump=2)),
    "counting_p2": (dict(n=12000, contigs=2, flavor="cbf", bfbytes=1 << 13), dict(min_threshold=2, max_threshold=5)),
    "snv_mode": (dict(n=6000, contigs=2, flavor="iupac", bfbytes=1 << 13), dict(snv=1)),
}


def main():
    out_root = os.path.join(HERE, "cases")
    for name, (ckw, pkw) in CASES.items():
        d = os.path.join(out_root, name)
        if os.path.isdir(d):
            shutil.rmtree(d)
        tmp = os.path.join("/tmp", "golden_" + name)
        case = H.make_case(tmp, 20251031, **ckw)
        os.makedirs(d)
        shutil.copy(case["draft"], os.path.join(d, "draft.fa"))
        shutil.copy(case["bf"], os.path.join(d, "filter.bf"))
        if case["rep"]:
            shutil.copy(case["rep"], os.path.join(d, "secondary.bf"))
        hp = H.default_params(**pkw)
        H.run_oracle(os.path.join(d, "draft.fa"), os.path.join(d, "filter.bf"), hp, os.path.join(d, "expected"),
                     os.path.join(d, "secondary.bf") if case["rep"] else None)
        # the VCF minus its date / input-path header lines
        with open(os.path.join(d, "expected_variants.vcf.body"), "w") as f:
            f.write("\n".join(H.vcf_body(os.path.join(d, "expected_variants.vcf"))) + "\n")
        os.remove(os.path.join(d, "expected_variants.vcf"))
        with open(os.path.join(d, "params.txt"), "w") as f:
            f.write(" ".join(H.oracle_args(hp)) + "\n")
        print(name, sum(1 for _ in open(os.path.join(d, "expected_changes.tsv"))) - 1, "rows")


if __name__ == "__main__":
    main()
