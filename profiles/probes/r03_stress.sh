# the chain tests with nearly every read sent through kernel D (a first-pass arena of 16 entries) and every eligible chain handed to the whole wave,
# and with entries as rows throughout: 71 + 43 tests green on the round's final tree
cd "$GRAFT_REPO_ROOT"
NABWA_CAP1=16 NABWA_DEEP_COOP=64 timeout -k 10 900 python -m pytest tests/test_gpu_pe.py tests/test_gpu_se.py tests/test_gpu_records.py tests/test_gpu_bam.py tests/test_gpu_poscache.py tests/test_gpu_aln_cli.py -q -m gpu 2>&1 | tail -3
NABWA_CAP1=16 NABWA_DEEP_COOP=1 NABWA_DEEP_KEYFORM=0 timeout -k 10 600 python -m pytest tests/test_gpu_pe.py tests/test_gpu_se.py tests/test_gpu_records.py -q -m gpu 2>&1 | tail -3
