#!/bin/bash
# One iteration of the bf16 training work on the GPU box: the bf16 parity tests, then cfg3-5 step times with the per-node profile.
#   gpurun --timeout 900 -- 'bash tools/gpu_bf16_iter.sh <tag> [pytest -k expression]'
TAG=${1:?tag}
KEXPR=${2:-}
export OUT=gpurun_out/$TAG TMPDIR=/tmp
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
(timeout -s KILL 700 python -m pytest tests/test_bf16_conv_gpu.py tests/test_bf16_wgrad_gpu.py tests/test_train_gpu.py tests/test_train_baseline_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider ${KEXPR:+-k "$KEXPR"} 2>&1 | tail -15) > $OUT/pytest.log
fi
(timeout -s KILL 400 python tools/train_bench.py --cfg ${CFGS:-3 4 5} --precision bf16 --steps ${STEPS:-8} --warmup 3 --profile-nodes 2>$OUT/nodes.log | cut -c1-330) > $OUT/train.log
tail -6 $OUT/pytest.log 2>/dev/null
python - <<PY
import json
for l in open("$OUT/train.log"):
    if l.startswith("{"):
        try:
            d = json.loads(l)
        except ValueError:
            import re
            m = re.search(r'"cfg": (\d).*?"ms_per_step": ([\d.]+)', l)
            print("cfg", m.group(1), "ms", m.group(2)) if m else None
            continue
        print("cfg", d["cfg"], "ms", d["ms_per_step"], "parity", (d.get("parity") or {}).get("ok"))
PY
sed -n '/per-phase totals/,/slowest entries/p' $OUT/nodes.log
