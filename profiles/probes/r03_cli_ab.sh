# nabwa_bam2bam file to file with the batch streams of both kinds (blocking streams serialise with the default stream), and the
# stage times of the records-in -> records-out leg
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/cli
for m in 1 0; do
  echo "== NABWA_STREAM_BLOCKING=$m, single-end"
  NABWA_STREAM_BLOCKING=$m timeout -k 10 400 python3 profiles/probes/cli_rate.py 8000000 /tmp/cli_rate > gpurun_out/cli/rate_se_$m.log 2>&1; echo rc=$?
  grep -a "wall\|search threads\|passes 1 and 2\|output records" gpurun_out/cli/rate_se_$m.log | cut -c1-330
  echo "== NABWA_STREAM_BLOCKING=$m, paired"
  NABWA_STREAM_BLOCKING=$m CLI_RATE_PAIRED=1 timeout -k 10 400 python3 profiles/probes/cli_rate.py 4000000 /tmp/cli_rate > gpurun_out/cli/rate_pe_$m.log 2>&1; echo rc=$?
  grep -a "wall\|search threads\|passes 1 and 2\|output records" gpurun_out/cli/rate_pe_$m.log | cut -c1-330
done
