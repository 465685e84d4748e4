for i in 1 2 3; do timeout 300 python tools/shard_test.py 2 3000 3500 12 400 cover 2>&1 | tail -3; done
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python tools/window_profile.py early 2000 200
bash tools/pmc_traffic_r02.sh > gpurun_out/r02q_pmc.log 2>&1; tail -60 gpurun_out/r02q_pmc.log
