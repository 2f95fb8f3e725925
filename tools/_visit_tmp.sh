cd "$GRAFT_REPO_ROOT"; T=r07d; O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
echo "== per-line cross check, verbose"
( time TFGPU_JSON_TILES=0 timeout 150 python -X faulthandler -m pytest tests/test_gpu_json.py -m gpu -v --tb=short -x -k "edge or random or flat or canon or rules or aux or messages or tile_path or float or rest" -o faulthandler_timeout=100 ) 2>&1 | tail -60
