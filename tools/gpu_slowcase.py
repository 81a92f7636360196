"""round 4: the slow fuzz case (seed 42424200091) through the library with NTEDIT_HIP_DEBUG, timing each variant"""
import os, sys, time, tempfile, filecmp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import ntedit_amd
H.build_oracle()
tmp = tempfile.mkdtemp()
case_kw = {'n': 42889, 'contigs': 3, 'k': 128, 'hashes': 2, 'p_sub': 0.01, 'p_ins': 0.005, 'p_del': 0.002, 'flavor': 'lower sec', 'bfbytes': 131072}
base = {'mode': 1, 'mask': 1, 'jump': 2, 'max_insertions': 5, 'max_deletions': 5, 'min_contig_len': 0, 'missing_threshold': 9.0, 'edit_threshold': 25.0, 'start_grid': 16, 'event_budget': 8}
case = H.make_case(tmp, 42424200091, **case_kw)
for name, par in (("m0 budget8", dict(base, mode=0, mask=0)),):
    pol = ntedit_amd.Polisher(0)
    pol.load_filter_file(case["bf"], 0)
    if case["rep"]:
        pol.load_filter_file(case["rep"], 1)
    pol.set_params(ntedit_amd.default_params(**par))
    t0 = time.time()
    st = pol.polish_records(H.read_fasta(case["draft"]), os.path.join(tmp, "g"))
    print("==", name, "%.1f s" % (time.time() - t0), "events", st.events, "applied", st.events_applied, "machine ms %.1f" % st.ms_machine, flush=True)
    pol.close()
