"""Multi-GPU layout of the hot path: one process per GPU (torch.distributed,
backend "nccl" = RCCL on ROCm, "gloo" on CPU for the tests).

The path shards embarrassingly: contigs are independent (ntedit.cpp:2213-2252
hands one contig to each OpenMP thread), so the only data-path collective is
ONE broadcast of the Bloom filter bit array from rank 0 at start-up; after that
ranks never talk until the host-side gather of per-shard outputs, which is
concatenated back in input order (= the reference at -t 1)."""
import os

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_contigs(lens, world, min_len=0):
    """Greedy LPT partition of contig indices by BASES (not by count): returns a list of
    index arrays, one per rank, each in input order.  Deterministic on every rank."""
    lens = np.asarray(lens, dtype=np.int64)
    idx = [i for i in range(len(lens)) if lens[i] >= min_len]
    order = sorted(idx, key=lambda i: (-int(lens[i]), i))
    load = [0] * world
    parts = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        parts[r].append(i)
        load[r] += int(lens[i])
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def broadcast_filter_tensor(t, src=0):
    """The path's single collective: broadcast the filter bit array (uint8 tensor, on the
    GPU for nccl/RCCL, on the CPU for gloo) from rank `src` to every rank."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def shared_filter(polisher, nbytes, hash_num, k, slot=0):
    """Allocate the filter bit array as a torch tensor on this rank's GPU and let the library
    adopt it (ntedit_hip_set_filter_device).  Every rank calls this with the same geometry;
    rank `src` then fills it (filter_insert / copy_) and broadcast_filter() ships it."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    nbytes = (int(nbytes) + 7) // 8 * 8
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    polisher.set_filter_device(buf.data_ptr(), nbytes, hash_num, k, slot=slot)
    if not hasattr(polisher, "_filter_keepalive"):
        polisher._filter_keepalive = {}
    polisher._filter_keepalive[slot] = buf
    return buf


def broadcast_filter(buf, src=0):
    """The path's single data-path collective: one broadcast of the filter bit array
    (RCCL over xGMI).  `buf` is the tensor returned by shared_filter()."""
    import torch
    torch.cuda.synchronize()
    broadcast_filter_tensor(buf, src)
    torch.cuda.synchronize()
    return buf


def load_and_broadcast_filter(polisher, path, src=0, slot=0):
    """Rank `src` reads a btllib .bf file; every rank ends up with the same filter in HBM."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    dev = torch.device("cuda", torch.cuda.current_device())
    hdr = torch.zeros(4, dtype=torch.int64, device=dev)
    data = None
    if rank == src:
        meta = {}
        with open(path, "rb") as f:
            first = f.readline()
            counting = b"Counting" in first
            while True:
                line = f.readline()
                if not line or line.startswith(b"[HeaderEnd]"):
                    break
                if b"=" in line:
                    key, val = [x.strip() for x in line.split(b"=", 1)]
                    meta[key.decode()] = val.decode().strip('"')
            off = f.tell()
        nbytes = int(meta["bytes"])
        data = np.memmap(path, dtype=np.uint8, mode="r", offset=off, shape=(nbytes,))
        hdr[0], hdr[1], hdr[2], hdr[3] = int(meta["k"]), int(meta["hash_num"]), nbytes, int(counting)
    broadcast_filter_tensor(hdr, src)
    k, h, nbytes, counting = [int(x) for x in hdr.tolist()]
    pad = (nbytes + 7) // 8 * 8
    buf = torch.zeros(pad, dtype=torch.uint8, device=dev)
    if rank == src:
        buf[:nbytes].copy_(torch.from_numpy(np.ascontiguousarray(data)))
    broadcast_filter(buf, src)
    polisher.set_filter_device(buf.data_ptr(), pad, h, k, slot=slot, counting=bool(counting))
    if not hasattr(polisher, "_filter_keepalive"):
        polisher._filter_keepalive = {}
    polisher._filter_keepalive[slot] = buf
    return buf


def run_sharded(records, polish_fn, out_prefix, min_len, rank, world, barrier=None):
    """records: [(header, seq)] known on every rank.  polish_fn(sub_records, prefix) writes
    <prefix>_edited.fa / <prefix>_changes.tsv for its contigs (each file WITHOUT any shared
    header handling: the TSV starts with the header line).  Rank 0 then gathers the per-rank
    files into <out_prefix>_* in input order."""
    lens = [len(s) for _, s in records]
    parts = shard_contigs(lens, world, min_len)
    mine = parts[rank]
    sub = [records[i] for i in mine]
    shard_prefix = "%s.shard%d" % (out_prefix, rank)
    polish_fn(sub, shard_prefix)
    if barrier:
        barrier()
    if rank == 0:
        merge_shards(records, parts, out_prefix, min_len)


def _split_fasta_records(path):
    recs = []
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline()
            recs.append(h + s)
    return recs


def merge_shards(records, parts, out_prefix, min_len):
    """Host-side gather: interleave the shard outputs back into input order.  _edited.fa has two
    lines per contig; _changes.tsv / _variants.vcf rows carry the contig header in column 1 and are
    contiguous per contig inside a shard."""
    world = len(parts)
    fa = [_split_fasta_records("%s.shard%d_edited.fa" % (out_prefix, r)) for r in range(world)]

    def load_rows(suffix, is_header):
        header, rows = None, []
        for r in range(world):
            path = "%s.shard%d%s" % (out_prefix, r, suffix)
            if not os.path.exists(path):
                return None, None
            with open(path, "rb") as f:
                lines = [l for l in f.read().split(b"\n") if l]
            hd = [l for l in lines if is_header(l)]
            if r == 0:
                header = hd
            rows.append([l for l in lines if not is_header(l)])
        return header, rows

    first_tsv = [True]

    def tsv_header(l):
        return l.startswith(b"ID\tbpPosition+1\t")

    tsv_hdr, tsv_rows = load_rows("_changes.tsv", tsv_header)
    vcf_hdr, vcf_rows = load_rows("_variants.vcf", lambda l: l.startswith(b"#"))
    owner = {}
    for r in range(world):
        for j, i in enumerate(parts[r]):
            owner[int(i)] = (r, j)
    cur_t = [0] * world
    cur_v = [0] * world
    ovcf = open(out_prefix + "_variants.vcf", "wb") if vcf_rows is not None else None
    with open(out_prefix + "_edited.fa", "wb") as ofa, open(out_prefix + "_changes.tsv", "wb") as otsv:
        for l in tsv_hdr:
            otsv.write(l + b"\n")
        if ovcf:
            for l in vcf_hdr:
                ovcf.write(l + b"\n")
        for i, (hdr, seq) in enumerate(records):
            if i not in owner:
                continue
            r, j = owner[i]
            ofa.write(fa[r][j])
            key = bytes(hdr) + b"\t"
            c = cur_t[r]
            while c < len(tsv_rows[r]) and tsv_rows[r][c].startswith(key):
                otsv.write(tsv_rows[r][c] + b"\n")
                c += 1
            cur_t[r] = c
            if ovcf:
                c = cur_v[r]
                while c < len(vcf_rows[r]) and vcf_rows[r][c].startswith(key):
                    ovcf.write(vcf_rows[r][c] + b"\n")
                    c += 1
                cur_v[r] = c
    if ovcf:
        ovcf.close()
    for r in range(world):
        for suffix in ("_edited.fa", "_changes.tsv", "_variants.vcf"):
            path = "%s.shard%d%s" % (out_prefix, r, suffix)
            if os.path.exists(path):
                os.remove(path)
