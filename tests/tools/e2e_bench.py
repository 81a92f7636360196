"""TEST INFRASTRUCTURE (runs the CPU oracle as the checker; lives under tests/ for that reason).
End-to-end region of SURVEY.md 8(d): the `ntedit` host binary from FASTA on disk to
_edited.fa/_changes.tsv on disk (reference: "reading/processing input sequence" ->
"process complete", ntedit.cpp:2589-2598), next to the kernel region it contains.
usage (GPU box): python tests/tools/e2e_bench.py [bases] [workdir]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import ntedit_amd  # noqa: E402
from ntedit_amd.synth import SyntheticJob  # noqa: E402


def main():
    bases = float(sys.argv[1]) if len(sys.argv) > 1 else 1e9
    work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/ntedit_e2e"
    os.makedirs(work, exist_ok=True)
    dev = torch.device("cuda", 0)
    pol = ntedit_amd.Polisher(0)
    pol.set_params(ntedit_amd.default_params())
    job = SyntheticJob(pol, bases, filter_bytes=1 << 32, device=dev)
    bf = os.path.join(work, "truth_k25.bf")
    pol.filter_save_file(bf)
    host = job.batch.cpu().numpy()
    draft = os.path.join(work, "draft.fa")
    with open(draft, "wb") as f:
        for i, (o, l) in enumerate(zip(job.offsets.tolist(), job.lens.tolist())):
            f.write(b">contig%d len=%d\n" % (i, l))
            f.write(host[o:o + l + 1].tobytes())  # sequence + '\n'
    pol.close()
    del job
    torch.cuda.empty_cache()
    out = {"bases": int(bases), "draft_bytes": os.path.getsize(draft)}
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "ntedit_amd", "ntedit"), "-f", draft, "-r", bf, "-b",
                        os.path.join(work, "gpu"), "--report"], capture_output=True, text=True)
    out["cli_wall_s"] = round(time.perf_counter() - t0, 3)
    assert r.returncode == 0, r.stderr
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    out["cli_polish_region_s"] = rep["seconds"]
    out["cli_gpu_ms"] = rep["gpu_ms"]
    out["cli_stage_seconds"] = {"read": rep["read_s"], "polish_call": rep["polish_call_s"], "write": rep["write_s"]}
    out["cli_mbases_per_s_polish_region"] = round(rep["bases"] / rep["seconds"] / 1e6, 1)
    out["events_applied"] = rep["events_applied"]
    # CPU oracle, same files, on the first ~30 Mbases (whole-file run would take minutes)
    small = os.path.join(work, "draft_small.fa")
    with open(draft, "rb") as fi, open(small, "wb") as fo:
        n = 0
        while n < 30e6:
            h = fi.readline()
            s = fi.readline()
            if not h:
                break
            fo.write(h)
            fo.write(s)
            n += len(s) - 1
    import helpers as H
    H.build_oracle()
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(H.ORACLE_BUILD, "ntedit_oracle"), "-f", small, "-r", bf, "-b",
                        os.path.join(work, "cpu"), "--report"], capture_output=True, text=True)
    out["oracle_wall_s_incl_filter_load"] = round(time.perf_counter() - t0, 3)
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    out["oracle_polish_region_s"] = rep["seconds"]
    out["oracle_mbases_per_s_polish_region"] = round(rep["bases"] / rep["seconds"] / 1e6, 2)
    # the GPU output restricted to the same contigs must equal the oracle's
    subprocess.run([os.path.join(ROOT, "ntedit_amd", "ntedit"), "-f", small, "-r", bf, "-b", os.path.join(work, "gpu_small")],
                   capture_output=True, text=True, check=True)
    same = all(open(os.path.join(work, "cpu" + s), "rb").read() == open(os.path.join(work, "gpu_small" + s), "rb").read()
               for s in ("_edited.fa", "_changes.tsv"))
    out["small_outputs_identical_to_oracle"] = same
    print(json.dumps(out))


if __name__ == "__main__":
    main()
