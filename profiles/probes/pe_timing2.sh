cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/t1
NABWA_TIMING=1 timeout -k 10 400 python3 bench.py --pe --pairs 1000000 --steps 2 --warmup 1 --no-cpu > gpurun_out/t1/pe.json 2> gpurun_out/t1/pe.err; echo rc=$?
python3 -c "import json;d=json.load(open('gpurun_out/t1/pe.json'));print('pe', d['value'], d['ms_per_step'], d['config']['stage_ms'])"
grep "pe_finish 1000000\|local_align" gpurun_out/t1/pe.err | tail -4 | cut -c1-300
