"""round 6: rehearsal of the multi-GPU DRIVER (python -m ntedit_amd.run) with N ranks on the one GPU of this box (gloo:
RCCL refuses two ranks on one device): a synthetic draft + filter on disk, the driver at N = 1 (RCCL) and N = 2, 4, 8
(gloo), outputs compared byte for byte with the N = 1 run, every rank's --report line kept -- host seconds per rank for the
index, the plan, the polish calls and the gather, bytes of the draft a rank read.  The GPU times mean nothing (N ranks
share one device); the HOST side is what this measures.
usage: python tools/gpu_rehearse_run.py [bases] [out.json]"""
import filecmp
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntedit_amd  # noqa: E402
from ntedit_amd.synth import SyntheticJob  # noqa: E402


def main():
    bases = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0e9
    out_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r6_rehearsal_run.json"
    work = "/tmp/ntedit_reh"
    os.makedirs(work, exist_ok=True)
    pol = ntedit_amd.Polisher(0)
    pol.set_params(ntedit_amd.default_params())
    job = SyntheticJob(pol, bases, filter_bytes=1 << 31, device=torch.device("cuda", 0))
    bf = os.path.join(work, "truth_k25.bf")
    pol.filter_save_file(bf)
    host = job.batch.cpu().numpy()
    draft = os.path.join(work, "draft.fa")
    with open(draft, "wb") as f:
        for i, (o, l) in enumerate(zip(job.offsets.tolist(), job.lens.tolist())):
            f.write(b">contig%d len=%d\n" % (i, l))
            seq = host[o:o + l]
            for a in range(0, l, 6000000):  # (lines of 60 bases, written in stretches)
                part = seq[a:a + 6000000]
                n = len(part) // 60 * 60
                body = part[:n].reshape(-1, 60)
                import numpy as np
                nl = np.full((body.shape[0], 1), 10, dtype=np.uint8)
                f.write(np.concatenate([body, nl], axis=1).tobytes())
                if n < len(part):
                    f.write(part[n:].tobytes() + b"\n")
    pol.close()
    del job, host
    torch.cuda.empty_cache()
    res = {"bases": int(bases), "draft_file_bytes": os.path.getsize(draft), "note": __doc__.split("usage")[0].strip(), "runs": []}
    for n in (1, 2, 4, 8):
        pre = os.path.join(work, "out%d" % n)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(29700 + n), "-m", "ntedit_amd.run", "-f", draft, "-r", bf, "-b", pre, "--report"]
        if n > 1:
            cmd += ["--backend", "gloo"]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            res["runs"].append({"ranks": n, "error": r.stderr[-1500:]})
            continue
        reports = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"rank"')), key=lambda x: x["rank"])
        same = None
        if n > 1:
            same = all(filecmp.cmp(pre + s, os.path.join(work, "out1") + s, shallow=False)
                       for s in ("_edited.fa", "_changes.tsv"))
        res["runs"].append({"ranks": n, "backend": "nccl (RCCL)" if n == 1 else "gloo, all ranks on ONE GPU", "wall_s": round(wall, 2),
                            "identical_to_one_rank": same, "per_rank": reports})
        print("ranks %d: wall %.1f s, identical %s, run_s %s, phases %s, read MB %s" % (
            n, wall, same, [x["run_s"] for x in reports], [x["phases_s"] for x in reports],
            [round(x["draft_bytes_read"] / 1e6) for x in reports]), flush=True)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
