#!/bin/bash
# debugging visit: the whole -m gpu suite with RCCL warnings on, full log kept
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/dbg; mkdir -p $OUT
NCCL_DEBUG=WARN timeout 900 python -m pytest tests -m gpu -q --tb=long 2>&1 | grep -v "Could not read node" > $OUT/full.log; tail -3 $OUT/full.log
