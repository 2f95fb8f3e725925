"""Hash-partition exchange (BASELINE.json configs[4]: debezium stream → hash-partition → dedup → Kafka sink, 8×MI355X):
the one step of the path that moves rows BETWEEN GPUs.

Local half (device): `sharder_transformer` with shardsCount = world size writes PartID = CRC32_IEEE(join(
SerializeToString(cols), ".")) % world (pkg/transformer/registry/sharder/sharder.go:130-145) and `tfgpu_partition`
regroups the rows by it, so every column buffer is one contiguous run per destination rank.

Exchange (this module): ONE `all_to_all_single` per column buffer — values, string lengths, string bytes, validity —
over torch.distributed: backend "nccl" is RCCL on ROCm, whose all-to-all drives each GPU's 7 xGMI links at once (a ring
would be per-link bound); "gloo" on CPU tensors in the tests.  Rows arrive grouped by source rank, each group in its
original order, i.e. the order a single process would have produced for that key range.  No all-reduce anywhere.

PyTorch is plumbing here (the collective and the tensor views over the library's buffers); the row data is produced
and consumed by the HIP kernels behind the C ABI.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import abi


def _a2a(dist, t_in, in_splits: Sequence[int], out_splits: Sequence[int], elem: int = 1):
    """all_to_all_single of a 1-D tensor whose segment d (in_splits[d] * elem entries) goes to rank d."""
    import torch
    out = torch.empty(int(sum(out_splits)) * elem, dtype=t_in.dtype, device=t_in.device)
    dist.all_to_all_single(out, t_in.contiguous(), [int(s) * elem for s in out_splits], [int(s) * elem for s in in_splits])
    return out


def exchange_counts(dist, counts: Sequence[int], device="cpu") -> List[int]:
    """counts[d] = rows this rank sends to rank d → rows this rank receives from each rank."""
    import torch
    t = torch.tensor(list(counts), dtype=torch.int64, device=device)
    out = torch.empty_like(t)
    dist.all_to_all_single(out, t)
    return [int(x) for x in out.tolist()]


def exchange_host_batch(dist, b: abi.Batch, counts: Sequence[int]) -> abi.Batch:
    """The exchange over a HOST batch (numpy buffers, CPU tensors): what the gloo tests run, and the reference
    semantics of the device path — `b` is already grouped by destination with `counts` rows per rank."""
    import torch
    world = dist.get_world_size()
    assert len(counts) == world and sum(counts) == b.nrows
    recv = exchange_counts(dist, counts)
    n_out = sum(recv)
    bounds = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)

    def xchg(c):
        o = abi.Column(c.name, c.dtype, c.repr)
        if c.repr in abi.VAR_REPRS:
            lens = np.diff(c.offsets.astype(np.int64)).astype(np.uint32)
            rl = _a2a(dist, torch.from_numpy(lens.view(np.int32)), counts, recv).numpy().view(np.uint32)
            # bytes per destination = sum of the lengths of that run
            bsend = [int(c.offsets[bounds[d + 1]]) - int(c.offsets[bounds[d]]) for d in range(world)]
            brecv = exchange_counts(dist, bsend)
            data = np.ascontiguousarray(c.data[: int(c.offsets[-1])], dtype=np.uint8)
            rd = _a2a(dist, torch.from_numpy(data), bsend, brecv).numpy()
            o.offsets = np.concatenate([[0], np.cumsum(rl, dtype=np.uint64)]).astype(np.uint32)
            o.data = rd
        else:
            w = np.dtype(abi.REPR_NP[c.repr]).itemsize
            raw = np.ascontiguousarray(c.values, dtype=abi.REPR_NP[c.repr]).view(np.uint8)
            o.values = _a2a(dist, torch.from_numpy(raw), counts, recv, w).numpy().view(abi.REPR_NP[c.repr])
            if c.nanos is not None:
                o.nanos = _a2a(dist, torch.from_numpy(np.ascontiguousarray(c.nanos, dtype=np.int32)), counts, recv).numpy()
        if c.validity is not None:
            v = np.ascontiguousarray(c.validity, dtype=np.uint8)
            o.validity = _a2a(dist, torch.from_numpy(v), counts, recv).numpy().astype(bool)
        return o
    cols = [xchg(c) for c in b.cols]
    out = abi.Batch(cols, n_out, b.table_ns, b.table_name)
    if b.kind is not None:
        out.kind = _a2a(dist, torch.from_numpy(np.ascontiguousarray(b.kind, dtype=np.uint8)), counts, recv).numpy()
    if b.src_row is not None:
        out.src_row = _a2a(dist, torch.from_numpy(np.ascontiguousarray(b.src_row, dtype=np.int32)), counts, recv).numpy()
    old_keys = list(getattr(b, "old_keys", None) or [])
    if old_keys:  # ChangeItem.OldKeys travel with their rows: Collapse on the receiving rank reads them
        out.old_keys = [xchg(c) for c in old_keys]
        pres = getattr(b, "old_present", None)
        pres = np.ones(b.nrows, np.uint8) if pres is None else np.ascontiguousarray(pres, dtype=np.uint8)
        out.old_present = _a2a(dist, torch.from_numpy(pres), counts, recv).numpy().astype(bool)
    out.part_id = np.full(n_out, dist.get_rank(), dtype=np.uint32)
    out.recv_counts = recv  # rows received from each source rank, in rank order
    if getattr(b, "schema", None) is not None:
        out.schema = b.schema  # TableSchema (PrimaryKey flags): constant per table, never exchanged
    return out


def exchange_device_batch(dist, lib, db, counts: Sequence[int], schema: "abi.Schema" = None):
    """The same exchange with the column buffers left in HBM: torch tensors view the library's device buffers (CUDA
    array interface), RCCL moves them, and the received buffers become a library batch again (device-to-device)."""
    import ctypes as C

    import torch
    world = dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    recv = exchange_counts(dist, counts, dev)
    n_in, n_out = int(sum(counts)), int(sum(recv))
    v = db.view()

    class _View:  # a device pointer as a CUDA-array-interface object
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (max(int(nbytes), 1),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    def tensor(ptr, nbytes):
        return torch.as_tensor(_View(ptr, nbytes), device=dev)[: int(nbytes)]

    lib.synchronize()  # the partition kernels ran on the library's stream, the collective runs on torch's
    nold = int(v.n_old_keys)
    keep, carr = [], (abi.CColumn * max(v.ncols + nold, 1))()
    bounds = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)

    def xchg_bits(ptr):
        """a bitmap is not byte-aligned per destination: exchange one byte per row, repack"""
        bits = tensor(ptr, (n_in + 7) // 8)
        un = ((bits[:, None] >> torch.arange(8, device=dev, dtype=torch.uint8)) & 1).reshape(-1)[:n_in].contiguous()
        rb = _a2a(dist, un, counts, recv)
        pad = torch.zeros((n_out + 7) // 8 * 8, dtype=torch.uint8, device=dev)
        pad[:n_out] = rb
        packed = (pad.reshape(-1, 8) << torch.arange(8, device=dev, dtype=torch.uint8)).sum(1).to(torch.uint8)
        keep.append(packed)
        return packed.data_ptr()

    def xchg(c, dst):
        dst.name, dst.dtype, dst.repr = c.name, c.dtype, c.repr
        if c.repr in abi.VAR_REPRS:
            off = tensor(c.offsets, (n_in + 1) * 4).view(torch.int32)
            lens = (off[1:] - off[:-1]).contiguous()
            rl = _a2a(dist, lens, counts, recv)
            ob = off[torch.as_tensor(bounds, device=dev)].to(torch.int64)
            bsend = [int(x) for x in (ob[1:] - ob[:-1]).tolist()]
            brecv = exchange_counts(dist, bsend, dev)
            rd = _a2a(dist, tensor(c.data, int(c.data_len)), bsend, brecv)
            ro = torch.zeros(n_out + 1, dtype=torch.int32, device=dev)
            ro[1:] = torch.cumsum(rl, 0)
            keep.extend([rl, rd, ro])
            dst.offsets, dst.data, dst.data_len = ro.data_ptr(), rd.data_ptr() if rd.numel() else ro.data_ptr(), int(sum(brecv))
        else:
            w = np.dtype(abi.REPR_NP[c.repr]).itemsize
            rv = _a2a(dist, tensor(c.values, n_in * w), counts, recv, w)
            keep.append(rv)
            dst.values = rv.data_ptr()
            if c.nanos:
                rn = _a2a(dist, tensor(c.nanos, n_in * 4), counts, recv, 4)
                keep.append(rn)
                dst.nanos = rn.data_ptr()
        if c.validity:
            dst.validity = xchg_bits(c.validity)
    for i in range(v.ncols):
        xchg(v.cols[i], carr[i])
    for k in range(nold):
        xchg(v.old_keys[k], carr[v.ncols + k])
    hb = abi.CBatch()
    hb.nrows, hb.ncols, hb.cols, hb.mem = n_out, v.ncols, carr, abi.MEM_DEVICE
    hb.table_ns, hb.table_name = v.table_ns, v.table_name
    if nold:  # ChangeItem.OldKeys travel with their rows: tfgpu_collapse on the receiving rank reads them
        hb.n_old_keys = nold
        hb.old_keys = C.cast(C.byref(carr, C.sizeof(abi.CColumn) * v.ncols), C.POINTER(abi.CColumn))
        if v.old_keys_present:
            hb.old_keys_present = xchg_bits(v.old_keys_present)
    if v.kind:
        rk = _a2a(dist, tensor(v.kind, n_in), counts, recv)
        keep.append(rk); hb.kind = rk.data_ptr()
    if v.src_row:
        rs = _a2a(dist, tensor(v.src_row, n_in * 4), counts, recv, 4)
        keep.append(rs); hb.src_row = rs.data_ptr()
    if schema is not None:  # TableSchema (PrimaryKey flags) for tfgpu_collapse on the receiving rank: constant per table
        cs = schema.to_c()
        keep.append(cs)
        hb.schema = C.pointer(cs)
    torch.cuda.synchronize()
    h = C.c_void_p()
    lib._check(lib.load().tfgpu_batch_upload(C.byref(hb), C.byref(h)))
    return lib.DeviceBatch(h), recv
