cd "$GRAFT_REPO_ROOT"
bash profiles/probes/headline_quick.sh
NABWA_LIB=$PWD/gpurun_ab/lib_w4.so bash profiles/probes/headline_quick.sh
