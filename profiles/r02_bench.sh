#!/bin/bash
# The round's bench lines WITHOUT a profiler attached (rocprofv3's tracing slows the host-side legs; kernel statistics come from
# profiles/r02_collect.sh): bash profiles/r02_bench.sh  -> gpurun_out/r02b/*.json
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r02b && O=gpurun_out/r02b
run() { name=$1; shift; timeout -k 10 600 python3 bench.py "$@" > $O/$name.json 2> $O/$name.log; echo "$name rc=$?"; }
run headline --steps 5 --warmup 1 || exit 1
run repeats --repeats --steps 3 --warmup 1 || exit 1
run adna --adna --reads 1000000 --steps 3 --warmup 1 || exit 1
run adna_6m --adna --reads 6250000 --steps 2 --warmup 1 --no-e2e || exit 1      # config 5's share of one GPU: 50 M reads over 8
run pe --pe --pairs 1000000 --steps 3 --warmup 1 || exit 1
# the index in the reference's own 2 GB-per-direction form: no k-mer table, no packed text
NABWA_KMER_T=0 NABWA_TEXT_MODE=0 run headline_plain_index --steps 3 --warmup 1 --no-e2e || exit 1
ls -la $O
