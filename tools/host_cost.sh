P='import json,sys; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["config"]["host_cores_busy_per_rank"])'
python bench.py --no-cpu-baseline --no-extras --workers 1 --steps 6 --warmup 3 | python -c "$P" w1_default
RGBD_SPIN_WAIT=1 python bench.py --no-cpu-baseline --no-extras --workers 1 --steps 6 --warmup 3 | python -c "$P" w1_spin
RGBD_BLOCKING_SYNC=1 python bench.py --no-cpu-baseline --no-extras --workers 1 --steps 6 --warmup 3 | python -c "$P" w1_devblocking
RGBD_BLOCKING_SYNC=1 python bench.py --no-cpu-baseline --no-extras | python -c "$P" w16_devblocking
RGBD_BLOCKING_SYNC=1 RGBD_NO_GRAPH=1 python bench.py --no-cpu-baseline --no-extras | python -c "$P" w16_devblocking_eager
