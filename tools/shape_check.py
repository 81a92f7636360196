import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import ntedit_amd
from ntedit_amd.synth import SyntheticJob
pol = ntedit_amd.Polisher(0)
pol.set_params(ntedit_amd.default_params())
# filter from genome A (seed 1), draft = unrelated genome B: every k-mer absent (up to the FPR)
jobA = SyntheticJob(pol, 3e8, filter_bytes=1 << 30, seed=1)
del jobA
jobB = SyntheticJob(pol, 3e8, filter_bytes=1 << 30, seed=2, build_filter=False, mutate=False, n_runs=False)
for it in range(2):
    t = time.time(); res = pol.polish_batch(None, jobB.offsets, jobB.lens, device_ptr=jobB.device_ptr, n=jobB.n_bytes); dt = time.time() - t
    st = res.stats(); res.free()
    print("unrelated draft: %.0f Mbases, absent %d, events %d, %.3f s (gpu %.1f ms)" % (jobB.n_bases / 1e6, st.absent_kmers, st.events, dt, st.ms_total), flush=True)
# one giant contig of 1 Gbp
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
L = 1_000_000_000
codes = torch.randint(0, 4, (L,), dtype=torch.uint8, device="cuda", generator=gen)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
big = torch.cat([lut[codes.long()] if False else torch.take(lut, codes.long()), torch.tensor([10], dtype=torch.uint8, device="cuda")])
del codes
pol.filter_alloc(1 << 32, 3, 25)
pol.filter_insert(None, device_ptr=big.data_ptr(), n=big.numel())
# mutate 0.1 %
idx = torch.randint(0, L, (L // 1000,), device="cuda", generator=gen)
big[idx] = lut[torch.randint(0, 4, (idx.numel(),), device="cuda", generator=gen)]
torch.cuda.synchronize()
offs = np.array([0], dtype=np.uint64); lens = np.array([L], dtype=np.uint32)
for it in range(2):
    t = time.time(); res = pol.polish_batch(None, offs, lens, device_ptr=big.data_ptr(), n=big.numel()); dt = time.time() - t
    st = res.stats()
    print("one 1 Gbp contig: absent %d, events %d, %.3f s (gpu %.1f ms)" % (st.absent_kmers, st.events, dt, st.ms_total), flush=True)
host = big.cpu().numpy()
t = time.time(); res.write(host, offs, lens, [b"giant"], "/tmp/giant_edited.fa", "/tmp/giant_changes.tsv"); print("render+write %.2f s, subs %d" % (time.time() - t, res.stats().substitutions))
