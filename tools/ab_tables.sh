#!/bin/bash
# Same-box A/B of launch tables through the default bench (alternating runs): usage: bash tools/ab_tables.sh "<bench args>" table1.json table2.json ...
# ("" = the committed table).  Kernel decisions rest on this, not on per-layer serial timings: with four batches in flight a launch
# that is faster alone is not necessarily faster beside the other three batches' launches (profiles/r05/k_*).
ARGS=$1; shift
for i in $(seq 1 ${REPS:-3}); do for t in "$@"; do
  W2L_TUNE_TABLE=$t timeout 200 python bench.py --no-cpu-baseline --no-train-configs --steps 20 --warmup 3 --sustained-seconds 1.5 $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('table=%-55s value %.0f sustained %.0f serial %.3f ms' % ('$t' or 'committed', d['value'], d['windows']['sustained_value'], d['roofline']['serial_ms_per_step']))"
done; done
