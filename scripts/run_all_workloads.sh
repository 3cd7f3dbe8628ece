# every bench workload once; results -> gpurun_out/workloads_r06.jsonl
out=gpurun_out/workloads_r06.jsonl; : > $out
for w in orbit512 orbit256 crabwalk512 farwall768; do
  steps=200; [ $w = farwall768 ] && steps=40
  timeout 600 python bench.py --workload $w --steps $steps --warmup 10 --cpu-frames 6 --no-stress 2>gpurun_out/$w.err | tail -1 >> $out || echo "{\"workload\": \"$w\", \"failed\": true}" >> $out
  tail -2 gpurun_out/$w.err
done
python - <<'PY'
import json
for l in open('gpurun_out/workloads_r06.jsonl'):
    try: d=json.loads(l)
    except Exception as e: print('bad line', l[:200]); continue
    if 'value' not in d: print(d); continue
    print(d['config']['workload'][:40], round(d['value'],1), 'fps', d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['U_voxels_updated'], d.get('cpu_baseline',{}).get('value'), d['config']['pose_err_m_at_end'])
PY
