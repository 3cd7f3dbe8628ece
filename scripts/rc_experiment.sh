for e in 1; do
KT_EXTRA_FLAGS=-DKT_RC_EXPERIMENT=$e python -c "
from kintinuous_amd import build; build.build(force=True)"
python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('exp', $e, d['value'], d['stage_ms'])"
done
python -c "
from kintinuous_amd import build; build.build(force=True)"
