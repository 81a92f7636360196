"""Worker for test_dist_cpu.py: exercises the N>1 plumbing of ntedit_amd.dist on CPU (gloo): the single filter
broadcast, the by-bases partition into pieces (contigs cut into segments at event-free boundaries), the
verification / joined re-run of segments and the host-side gather by index.  The per-shard compute stand-in is the
test-only host build of the event machine + the product's renderer (the real ranks call the HIP library through
ntedit_amd.run.HipBackend, same interface)."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers as H  # noqa: E402
from ntedit_amd import dist as ndist  # noqa: E402
from ntedit_amd._lib import Segment  # noqa: E402


class HostsimBackend:
    """screen() / polish() of dist.run_sharded on the host simulation"""

    def __init__(self, bf, hp, blind=False):
        self.bf, self.hp, self.blind = bf, hp, blind
        self.reruns = 0

    def screen(self, blob):
        if self.blind:  # (tests: pretend every k-mer is in the filter -> cuts land anywhere)
            return np.zeros((len(blob) + 63) // 64, dtype=np.uint64)
        return H.oracle_screen(blob, self.bf)

    def _run(self, entries, segs, prefix, covers=None, sizes=None, cuts_ok=None):
        lib = H.hostsim_lib()
        lib.hostsim_set_cuts_ok(cuts_ok.ctypes.data_as(ctypes.c_void_p) if cuts_ok is not None else None)
        arr = (Segment * max(len(entries), 1))()
        for i, (off, halo, flags) in enumerate(segs):
            arr[i].pos_offset, arr[i].halo, arr[i].flags = off, halo, flags
        lib.hostsim_set_render_extras(arr, sizes.ctypes.data_as(ctypes.c_void_p) if sizes is not None else None,
                                      covers.ctypes.data_as(ctypes.c_void_p) if covers is not None else None,
                                      ctypes.c_uint(1), None)
        hp = H.default_params(**{f[0]: getattr(self.hp, f[0]) for f in self.hp._fields_})
        hp.min_contig_len = 0
        if prefix is None:
            blob, offs, lens, names = H.pack_batch([(e[0], bytes(e[1])) for e in entries], 0)
            n = len(names)
            rc = lib.hostsim_polish(
                ctypes.c_char_p(blob), ctypes.c_uint64(len(blob)), offs.ctypes.data_as(ctypes.c_void_p),
                lens.ctypes.data_as(ctypes.c_void_p), (ctypes.c_char_p * max(n, 1))(*names), ctypes.c_uint32(n),
                self.bf["data"].ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(self.bf["bytes"]),
                ctypes.c_uint32(self.bf["hash_num"]), ctypes.c_uint32(self.bf["k"]), None, ctypes.c_uint64(0),
                ctypes.c_uint32(0), ctypes.byref(hp), None, None, None, None, ctypes.c_int(0), ctypes.c_int(0), None, None)
            return rc
        rc, _, _ = H.run_hostsim([(e[0], bytes(e[1])) for e in entries], self.bf, hp, prefix)
        return rc

    def polish(self, entries, fa, tsv, vcf, append):
        n = len(entries)
        segs = [e[2] for e in entries]
        covers = np.zeros(max(n, 1), dtype=np.uint32)
        ok = np.ones(max(n, 1), dtype=np.uint8)
        # (no output: the renderer's verdict -- -7 = some cut is bad -- and the predicate the product's driver asks
        # before it writes anything, ntedit_hip_result_cuts_ok; they must agree)
        rc = self._run(entries, segs, None, covers=covers, cuts_ok=ok)
        assert rc in (0, -7), rc
        bad = [i for i in range(n) if not ok[i]]
        assert (rc == -7) == bool(bad), (rc, bad)
        # every cut the run ends behind is bad (the cover end alone is the weaker test the driver used to apply)
        weak = [i for i, (_, seq, (off, halo, flags)) in enumerate(entries) if halo and covers[i] > len(seq) - halo]
        assert set(weak) <= set(bad), (weak, bad)
        self.reruns += len(bad)
        segs = [(o, h, f | (ndist.SEG_SKIP if i in bad else 0)) for i, (o, h, f) in enumerate(segs)]
        sizes = np.zeros((max(n, 1), 3), dtype=np.uint64)
        tmp = fa + ".part"
        rc = self._run(entries, segs, tmp, sizes=sizes)
        assert rc == 0, rc
        for src, dst in ((tmp + "_edited.fa", fa), (tmp + "_changes.tsv", tsv), (tmp + "_variants.vcf", vcf)):
            with open(src, "rb") as fi, open(dst, "ab" if append else "wb") as fo:
                fo.write(fi.read())
            os.remove(src)
        return bad, sizes[:n]


def main():
    draft, bf_path, out_prefix = sys.argv[1:4]
    seg_bases = int(sys.argv[4]) if len(sys.argv) > 4 else None
    blind = len(sys.argv) > 5 and sys.argv[5] == "blind"
    rank, world, _ = ndist.init_process_group("gloo")
    # rank 0 owns the filter file; everybody else receives it through the one broadcast
    if rank == 0:
        bf = H.load_bf(bf_path)
        hdr = torch.tensor([bf["k"], bf["hash_num"], bf["bytes"]], dtype=torch.int64)
    else:
        bf = None
        hdr = torch.zeros(3, dtype=torch.int64)
    ndist.broadcast_filter_tensor(hdr, 0)
    k, h, nbytes = [int(x) for x in hdr]
    bits = torch.from_numpy(bf["data"].copy()) if rank == 0 else torch.empty(nbytes, dtype=torch.uint8)
    ndist.broadcast_filter_tensor(bits, 0)
    bf = {"k": k, "hash_num": h, "bytes": nbytes, "data": bits.numpy(), "counting": False}
    # identical bits everywhere?
    chk = torch.tensor([int(bits.to(torch.int64).sum())], dtype=torch.int64)
    lo = chk.clone()
    hi = chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert int(lo) == int(hi) == int(chk)

    # DIST_LAZY=1: the draft as the real driver takes it -- index only, bases read on demand (ntedit_amd.run.Draft over
    # ntedit_hip_fasta_open / _read); else the whole draft in memory
    lazy = None
    if os.environ.get("DIST_LAZY") == "1":
        from ntedit_amd.run import Draft
        lazy = Draft(draft)
        recs = lazy.records()
    else:
        recs = H.read_fasta(draft)
    hp = H.default_params()
    backend = HostsimBackend(bf, hp, blind)

    def write_headers(pre):
        open(pre + "_edited.fa", "wb").close()
        lib = H.oracle_lib()  # (header lines only: the shard bodies come from the product's renderer)
        with open(pre + "_changes.tsv", "w") as f:
            f.write("ID\tbpPosition+1\tOriginalBase\tNewBase\tSupport %d-mer (out of %g)\tAlt.Base1\tAlt.Support1\t"
                    "Alt.Base2\tAlt.Support2\tAlt.Base3\tAlt.Support3\n" % (k, -(-k // hp.jump)))
        open(pre + "_variants.vcf", "wb").close()

    halo = ndist.halo_bases(k, hp.max_insertions, hp.max_deletions)
    def all_gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    # DIST_GATHER=parallel: every rank copies its pieces to their places in the final files; else rank 0 merges
    mine = ndist.run_sharded(recs, backend, out_prefix, hp.min_contig_len, rank, world, k, halo, write_headers,
                             barrier=dist.barrier, seg_bases=seg_bases,
                             all_gather=all_gather if os.environ.get("DIST_GATHER") == "parallel" else None)
    if lazy is not None:
        with open("%s.read%d" % (out_prefix, rank), "w") as f:
            f.write("%d %d\n" % (lazy.bytes_read, sum(lazy.lens)))
        lazy.close()
    stats = torch.tensor([len(mine), sum(1 for p in mine if p.n_seg > 1), backend.reruns,
                          sum(p.end - p.start for p in mine)], dtype=torch.int64)
    allst = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allst, stats)
    if rank == 0:
        with open(out_prefix + ".stats", "w") as f:
            for s in allst:
                f.write(" ".join(str(int(x)) for x in s) + "\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
