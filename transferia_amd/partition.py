"""Hash-partition exchange (BASELINE.json configs[4]: debezium stream → hash-partition → dedup → Kafka sink, 8×MI355X):
the one step of the path that moves rows BETWEEN GPUs.

Local half (device): `sharder_transformer` with shardsCount = world size writes PartID = CRC32_IEEE(join(
SerializeToString(cols), ".")) % world (pkg/transformer/registry/sharder/sharder.go:130-145) and `tfgpu_partition`
regroups the rows by it, so every column buffer is one contiguous run per destination rank.

Exchange: `tfgpu_exchange` behind the C ABI (csrc/tf_exchange.hip) — ONE grouped RCCL send/recv over every column buffer
(values, text lengths, text bytes, validity, kinds, OldKeys), point-to-point so each GPU drives its 7 xGMI links at once
(a ring would be per-link bound).  `exchange_device_batch` is its wrapper: torch.distributed only carries the 128-byte
rendezvous id.  `exchange_host_batch` is the same exchange over host batches and a `gloo` group — the reference semantics
the CPU tests pin the device path to.  Rows arrive grouped by source rank, each group in its original order, i.e. the
order a single process would have produced for that key range.  No all-reduce anywhere.
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


_COMMS = {}


def device_comm(dist, lib):
    """The library communicator (tfgpu_comm_init) of a torch.distributed job: rank 0 makes the rendezvous id and the
    existing process group carries its 128 bytes — the role the coordinator's shared state plays for the Go workers.
    torch is only the control plane here; the payload moves inside tfgpu_exchange."""
    key = (id(dist), dist.get_rank(), dist.get_world_size())
    if key not in _COMMS:
        box = [lib.Comm.unique_id() if dist.get_rank() == 0 else None]
        if dist.get_world_size() > 1:
            dist.broadcast_object_list(box, src=0)
        _COMMS[key] = lib.Comm.create(box[0], dist.get_rank(), dist.get_world_size())
    return _COMMS[key]


def close_device_comms():
    for c in _COMMS.values():
        c.close()
    _COMMS.clear()


def exchange_device_batch(dist, lib, db, counts: Sequence[int], schema: "abi.Schema" = None):
    """The same exchange with the column buffers left in HBM, behind the C ABI: tfgpu_exchange (one grouped RCCL
    send/recv over all column buffers, csrc/tf_exchange.hip).  `schema` is accepted for symmetry with the host variant;
    the device batch already carries its TableSchema.  Returns (DeviceBatch, rows received per source rank)."""
    return device_comm(dist, lib).exchange(db, counts)
