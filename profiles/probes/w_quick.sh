cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
bash profiles/probes/headline_quick.sh
NABWA_LIB=$PWD/gpurun_ab/lib_wown.so bash profiles/probes/headline_quick.sh
done
