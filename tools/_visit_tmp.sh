cd "$GRAFT_REPO_ROOT"; T=r07e; O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --tb=short -x ) 2>&1 | tail -15 | tee $O/pytest_gpu.log
