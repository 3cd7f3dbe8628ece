timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('early throttle', round(d['value'],1), d['host_ms_per_frame'])"
KT_NO_THROTTLE=1 timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('early plain', round(d['value'],1), d['host_ms_per_frame'])"
done
