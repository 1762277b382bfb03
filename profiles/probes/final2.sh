cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/final && timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final/pytest.log 2>&1; echo suite rc=$?; tail -3 gpurun_out/final/pytest.log | cut -c1-220
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; echo smoke rc=$?; tail -2 gpurun_out/final/smoke.log | cut -c1-300
timeout -k 10 600 python3 bench.py --pe --pairs 1000000 --steps 3 --warmup 1 > gpurun_out/final/pe.json 2> gpurun_out/final/pe.log; echo pe rc=$?
