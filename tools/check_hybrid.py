import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ntedit_amd
from ntedit_amd.synth import SyntheticJob
dev = torch.device("cuda", 0)
pol = ntedit_amd.Polisher(0)
pol.set_params(ntedit_amd.default_params())
job = SyntheticJob(pol, float(sys.argv[1]) if len(sys.argv) > 1 else 3e8, filter_bytes=1 << 30, device=dev)
nw = (job.n_bytes + 63) // 64 + 1
outs = {}
for name, mode, frac in (("direct", 1, None), ("binned", 2, "1.0"), ("hybrid", 2, "0.6")):
    if frac is None:
        os.environ.pop("NTEDIT_HIP_HYBRID_FRAC", None)
    else:
        os.environ["NTEDIT_HIP_HYBRID_FRAC"] = frac
    pol.set_params(ntedit_amd.default_params(screen_mode=mode))
    bm = torch.full((nw,), -1, dtype=torch.int64, device=dev)
    for _ in range(2):
        ms = pol.screen_device(job.device_ptr, job.n_bytes, bm.data_ptr())
    torch.cuda.synchronize()
    outs[name] = bm[: nw - 1].clone()
    print(name, "ms", round(ms, 3), "set bits", int(sum(bin(int(x) & (2**64 - 1)).count("1") for x in outs[name][:2000].tolist())), flush=True)
for name in ("binned", "hybrid"):
    same = torch.equal(outs[name], outs["direct"])
    print(name, "== direct:", same)
    if not same:
        d = (outs[name] != outs["direct"]).nonzero().flatten()
        print("  differing words:", d.numel(), "first", d[:5].tolist(), "last", d[-5:].tolist(), "of", nw)
