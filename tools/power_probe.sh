#!/bin/bash
# Socket power / clocks sampled by rocm-smi while the default bench runs (GPU box): usage: bash tools/power_probe.sh [bench args]
# prints the samples and the bench line's value / sustained clock.  Evidence for "the four-batches-in-flight step is power-limited".
(timeout 200 python bench.py --no-cpu-baseline --no-train-configs --steps 20 --warmup 3 --sustained-seconds 6 "$@" 2>/dev/null > /tmp/pp_bench.json) &
BP=$!
sleep 9
for i in 1 2 3 4 5 6; do /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.7; done
wait $BP
python -c "import json; d=json.load(open('/tmp/pp_bench.json')); print('value', d['value'], 'sustained', d['windows']['sustained_value'], 'clock', d['roofline']['sustained_clock_mhz'], d['windows']['sustained']['clock_samples'])"
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i power
