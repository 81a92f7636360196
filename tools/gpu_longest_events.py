"""round 6: which events do the machine's launches wait for?  Profile build (libntedit_hip_prof.so: g_evlog), one class of
the genome-like structure at a time; prints the longest events of both kernels with the draft around their start.
usage: NTEDIT_HIP_LIB=.../libntedit_hip_prof.so NTEDIT_HIP_DEBUG=1 python tools/gpu_longest_events.py CLASS SHARE [BASES]"""
import json, os, re, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ntedit_amd
from ntedit_amd.synth import SyntheticJob

cls, share = sys.argv[1], float(sys.argv[2])
bases = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
fr = dict(simple=0, sat=0, dispersed=0, segdup=0, novel=0)
if cls == "genome":
    fr = None
elif cls != "iid":
    fr[cls] = share
pol = ntedit_amd.Polisher(0)
pol.set_params(ntedit_amd.default_params())
job = SyntheticJob(pol, bases, k=25, hash_num=3, filter_bytes=1 << 32, seed=20251031, draft_seed=20251032, structure="genome",
                   structure_fractions=fr)
pol.reserve(job.n_bytes, len(job.lens) + 64, on_device=1)
res = pol.polish_batch(None, job.offsets, job.lens, device_ptr=job.device_ptr, n=job.n_bytes)
res.free()
# second call: its stderr goes to a file
tmp = tempfile.NamedTemporaryFile(delete=False)
sys.stderr.flush()
keep = os.dup(2)
os.dup2(tmp.fileno(), 2)
res = pol.polish_batch(None, job.offsets, job.lens, device_ptr=job.device_ptr, n=job.n_bytes)
st = res.stats()
res.free()
os.dup2(keep, 2)
text = open(tmp.name).read()
print("== %s %.3f: machine %.2f ms, events %d, deferred %d" % (cls, share, st.ms_machine, st.events, st.events_deferred))
for line in text.splitlines():
    if "per-event launches" in line or "events " in line and "round A" in line:
        print(line[:600])
    m = re.search(r"(thread|wavefront)-per-event launches: .*failing\):(.*)", line)
    if not m:
        continue
    for ent in m.group(2).split(";")[:10]:
        ent = ent.strip()
        if not ent:
            continue
        pos = int(ent.split(":")[0])
        lo = max(0, pos - 60)
        ctx = bytes(job.batch[lo:pos + 240].cpu().numpy()).decode("latin1").replace("\n", "|")
        print("   %s  %s" % (m.group(1), ent))
        print("      ...%s[%s" % (ctx[:pos - lo], ctx[pos - lo:]))
# where the work is: gathers per 4 Mbase region of the batch (NTEDIT_HIP_REGIONS=1), and what the structure put there
import collections
reg = collections.Counter()
for line in text.splitlines():
    m = re.search(r"region (thread|wave) (\d+): (\d+) gathers", line)
    if m:
        reg[(m.group(1), int(m.group(2)))] += int(m.group(3))
if reg and job.structure is not None:
    feats = [(int(job.offsets[c]) + at, n, cl) for c, at, n, cl in job.structure.features]
    print("top regions (4 Mbase) by filter bytes gathered:")
    for (kind, r), v in reg.most_common(25):
        lo, hi = r << 22, (r + 1) << 22
        inside = collections.Counter()
        for at, n, cl in feats:
            ov = min(hi, at + n) - max(lo, at)
            if ov > 0:
                inside[cl] += ov
        print("   %-6s region %4d: %11d gathers; structure bases inside: %s" % (kind, r, v, dict(inside)))
    print("   all listed regions: thread %d, wave %d" % (sum(v for (k_, _), v in reg.items() if k_ == "thread"),
                                                         sum(v for (k_, _), v in reg.items() if k_ == "wave")))
pol.close()
