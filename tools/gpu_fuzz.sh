cd $GRAFT_REPO_ROOT
HWY_FUZZ_CHUNKS=40 timeout 1500 python -m pytest tests/test_fuzz_configs.py -m gpu -q -x 2>&1 | tail -15
