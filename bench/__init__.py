"""bench/ — the modules behind bench.py (the entry point at the repo root): common.py (constants, synthetic inputs, parity helpers),
cpu_workers.py (the cpu_baseline leg's host side), pipeline.py (pull || transform || push), wl_*.py (one module per BASELINE.json
configuration / component line), main.py (the driver contract: timed region, per-kernel events, the JSON line)."""
from .common import stage_shard, json_fields, make_messages  # noqa: F401  (tests/test_gpu_fullsize.py and tools/ablate_csv.py stage the bench's own shard)
