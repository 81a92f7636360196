"""Worker for test_dist_cpu.py: exercises the N>1 plumbing of ntedit_amd.dist on
CPU (gloo): the single filter broadcast, the by-bases contig sharding and the
host-side gather.  The per-shard compute stand-in is the test-only host build
of the event machine (the real ranks call the HIP library)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers as H  # noqa: E402
from ntedit_amd import dist as ndist  # noqa: E402


def main():
    draft, bf_path, out_prefix = sys.argv[1:4]
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

    recs = H.read_fasta(draft)
    hp = H.default_params()

    def polish_fn(sub, prefix):
        rc, _, _ = H.run_hostsim(sub, bf, hp, prefix)
        assert rc == 0

    ndist.run_sharded(recs, polish_fn, out_prefix, hp.min_contig_len, rank, world, barrier=dist.barrier)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
