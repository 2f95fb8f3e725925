#!/usr/bin/env python3
"""bench.py — the transform-stage hot path on ClickBench-hits-shaped batches, one MI355X per rank.

Default workload = BASELINE.json configs[1] ("ClickBench hits-1M CSV parse → mask(ip)+filter(EventDate) → devnull,
1×MI355X"): one *step* is `passes` passes of that path over one HBM-resident batch of 2^20 synthetic hits rows per GPU.
Rows shard by range across ranks with no data-path collective (weak scaling: every rank works on its own batch);
torch.distributed (nccl = RCCL) is used for the barrier and the max-over-ranks timing — and, in configs[4] only,
for the one real exchange of the path (hash-partition all-to-all).

    python bench.py [--workload csv|configs0|configs2|configs3|configs4|configs4d|json|sr|sr_proto|collapse|debezium|debezium_sr] --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (DESIGN.md "Measurement" explains every field).  Every workload reports `roofline` (dominant
kernel: algorithmic bytes per launch ÷ its HIP-event launch time, against the 8 TB/s HBM peak), `cpu_baseline` (the oracle — a C
restatement of the Go reference — on a bounded sample of the same input) and `parity` (the device's output for that sample against
the oracle's).  The default line also carries `other_configs`: configs 0 / 2 / 3 / 4 and the Debezium parser, each run as its own
short bench after the headline's timed region (rows/s, ms per pass, dominant kernel + fraction of the HBM peak, parity).

The code lives in bench/ (common.py, cpu_workers.py, pipeline.py, wl_*.py, main.py); this file is the entry point the driver runs.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bench.main import main  # noqa: E402  (bench/ the package: a directory wins over this file when `bench` is imported)

if __name__ == "__main__":
    main()
