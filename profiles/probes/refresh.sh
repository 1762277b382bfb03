cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r02c && O=gpurun_out/r02c
run() { name=$1; shift; timeout -k 10 600 python3 bench.py "$@" > $O/$name.json 2> $O/$name.log; echo "$name rc=$?"; }
run headline --steps 5 --warmup 1 || exit 1
run pe --pe --pairs 1000000 --steps 3 --warmup 1 || exit 1
