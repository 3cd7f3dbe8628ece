#!/bin/bash
# read-ahead stage, second cut: unrolled pyrDown taps, levels 2/3 + scaleDepth in one launch (KT_PREPARE_FUSED) -- parity, kernel times, bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c24; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_image.py tests/test_gpu_volume.py tests/test_gpu_configs.py tests/test_gpu_sweep.py tests/test_gpu_tracker.py tests/test_gpu_track.py tests/test_gpu_e2e.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
KT_PREPARE_FUSED=0 KT_PYR_FORM=1 timeout 900 python -m pytest tests/test_gpu_image.py tests/test_gpu_tracker.py -x -q -m gpu > $O/pytest_old.log 2>&1; echo "pytest old form rc $?"; tail -2 $O/pytest_old.log
i=0
for cfg in "KT_PREPARE_FUSED=0 KT_PYR_FORM=1" "KT_PREPARE_FUSED=0 KT_PYR_FORM=2" "KT_PREPARE_FUSED=1"; do
  i=$((i+1))
  ( cd /tmp && env $cfg rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-stress > $GRAFT_REPO_ROOT/$O/prof_$i.json 2> $GRAFT_REPO_ROOT/$O/prof_$i.err )
  python - <<PY
import sqlite3
db=sqlite3.connect("$O/prof_$i/p_results.db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; sym=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
for n,c,a,m in cur.execute(f"select s.kernel_name, count(*), avg(k.end-k.start), min(k.end-k.start) from {kd} k join {sym} s on k.kernel_id=s.id group by s.kernel_name"):
    if any(x in n for x in ("pyramid","scale_depth","tile_finish","bilateral2","prepare_fused")): print("$cfg |", n[:44].ljust(44), c, "avg %.1f min %.1f" % (a/1e3, m/1e3))
PY
done
for rep in 1 2; do i=0; for cfg in "KT_PREPARE_FUSED=0 KT_PYR_FORM=1" "KT_PREPARE_FUSED=1"; do
  i=$((i+1))
  env $cfg timeout 900 python bench.py --no-cpu-baseline > $O/bench_c${i}_$rep.json 2> $O/bench_c${i}_$rep.err; echo "cfg$i rep$rep rc $?"
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c24/bench_c*.json")):
    try: j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=j["roofline"]; s=j.get("roofline_stress") or {}
    print(f.split("/")[-1], "fps %.0f" % j["value"], "frac %.3f alone %.3f" % (r["frac"], r["frac_alone"] or 0), "serial", j.get("stage_ms"), "pipe", j.get("stage_ms_pipelined"),
          "| stress alone %.3f pipe %.3f frame %.3f / pipelined %.3f" % (s.get("frac_alone") or 0, s.get("frac_pipelined") or 0, s.get("frame_ms") or 0, s.get("frame_ms_pipelined") or 0), (s.get("stage_ms") or ""))
PY
