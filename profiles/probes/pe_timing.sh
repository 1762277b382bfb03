cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/pe_t
NABWA_TIMING=1 timeout -k 10 500 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu > gpurun_out/pe_t/out.json 2> gpurun_out/pe_t/err.log; echo rc=$?
python3 -c "import json;d=json.load(open('gpurun_out/pe_t/out.json'));print(d['value'], d['config']['stage_ms'], d['roofline']['search_kernel_ms'], d['roofline']['deep_kernel_ms'])"
grep -v "^\[synth\]" gpurun_out/pe_t/err.log | grep -i "pe_finish\|finish\|posn\|pairing\|rescue\|kernel D" | tail -24
