"""The multi-threaded oracle driver used by the full-size GPU parity tests (every contig of a 3 Gbp batch is
compared) must write exactly the files of the single-threaded oracle: contigs are polished concurrently and
longest first, the outputs are assembled in input order."""
import filecmp

import helpers as H


def _case(tmp_path, seed, **kw):
    case = H.make_case(str(tmp_path), seed, **kw)
    recs = H.read_fasta(case["draft"])
    return case, H.pack_batch(recs, 100)


def test_mt_files_equal_single_thread(tmp_path, oracle_build):
    case, (blob, offs, lens, names) = _case(tmp_path, 31337, contigs=9, n=12000, flavor="N lower")
    bf = H.load_bf(case["bf"])
    hp = H.default_params()
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"))
    for threads in (1, 4):
        pre = str(tmp_path / ("m%d" % threads))
        done = H.oracle_polish_flat_mt_files(blob, offs, lens, names, bf["data"], bf["hash_num"], bf["k"], threads,
                                             fa_path=pre + "_edited.fa", tsv_path=pre + "_changes.tsv",
                                             vcf_path=pre + "_body.vcf")
        assert done == int(lens.sum())
        assert filecmp.cmp(str(tmp_path / "o_edited.fa"), pre + "_edited.fa", shallow=False)
        assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), pre + "_changes.tsv", shallow=False)
        body = [l for l in open(str(tmp_path / "o_variants.vcf")).read().splitlines() if not l.startswith("#")]
        assert open(pre + "_body.vcf").read().splitlines() == body and len(body) > 10


def test_mt_files_secondary_and_params(tmp_path, oracle_build):
    import numpy as np
    case, (blob, offs, lens, names) = _case(tmp_path, 31338, contigs=5, n=15000, flavor="sec", k=35)
    bf, rep = H.load_bf(case["bf"]), H.load_bf(case["rep"])
    kw = dict(max_insertions=5, max_deletions=9)
    H.run_oracle(case["draft"], case["bf"], H.default_params(**kw), str(tmp_path / "o"), case["rep"])
    pre = str(tmp_path / "m")
    H.oracle_polish_flat_mt_files(np.frombuffer(blob, dtype=np.uint8), offs, lens, names, bf["data"], bf["hash_num"],
                                  bf["k"], 3, fa_path=pre + "_edited.fa", tsv_path=pre + "_changes.tsv",
                                  rep_bits=rep["data"], rep_hash_num=rep["hash_num"], **kw)
    assert filecmp.cmp(str(tmp_path / "o_edited.fa"), pre + "_edited.fa", shallow=False)
    assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), pre + "_changes.tsv", shallow=False)
