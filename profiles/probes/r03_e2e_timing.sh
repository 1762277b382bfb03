# stage timings of the records-in -> records-out leg (NABWA_TIMING prints of the library) at the headline size
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/e2e
NABWA_TIMING=1 timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 --no-cpu --extras off > gpurun_out/e2e/head.json 2> gpurun_out/e2e/head.err; echo rc=$?
python3 -c "import json;d=json.load(open('gpurun_out/e2e/head.json'));print('e2e', d['e2e']['reads_per_s'], d['e2e']['first_batch_reads_per_s'], d['e2e']['stage_ms'])"
grep -n "se_posn\|bam_batch\|se_refine\|refine_batch\|nabwa\] batch\|cal_sa_reg_gap\|upload\|sa_lookup" gpurun_out/e2e/head.err | tail -30 | cut -c1-400
