#!/bin/bash
# Copy the judged summaries of a GPU round from gpurun_out/ (scratch) into profiles/<round>/ (tracked).
# usage: [ROUND=r02] bash tools/collect_profiles.sh <inference tag, e.g. r01c> <training tag, e.g. t05> <dest prefix, e.g. c> [bf16 training tag]
ROUND=${ROUND:-r03}
INF=gpurun_out/$1; TR=gpurun_out/$2; P=profiles/$ROUND/$3
mkdir -p profiles/$ROUND
if [ -d "$INF" ]; then
  cp $INF/bench.json ${P}_bench.json
  [ -f $INF/bench_noremap.json ] && cp $INF/bench_noremap.json ${P}_bench_noremap.json
  cp $INF/layers.log ${P}_layers.log
  [ -f $INF/tune.json ] && cp $INF/tune.json ${P}_tune.json
  [ -f $INF/pytest_gpu.log ] && cp $INF/pytest_gpu.log ${P}_pytest_gpu.log
  [ -f $INF/smoke.log ] && cp $INF/smoke.log ${P}_smoke.log
  cp $INF/prof_stats/stats_kernel_stats.csv ${P}_kernel_stats.csv
  [ -f $INF/prof_stats_p1/stats_kernel_stats.csv ] && cp $INF/prof_stats_p1/stats_kernel_stats.csv ${P}_kernel_stats_pipeline1.csv
  names=(x sq lds fetch write)
  for i in 1 2 3 4; do
    f=$INF/prof_pmc$i/pmc${i}_counter_collection.csv
    [ -f $f ] && python tools/pmc_summary.py $f > ${P}_pmc${i}_${names[$i]}.txt
  done
  python - "$P" "$ROUND" "$INF" <<'PY'
import json, os, re, sys
p, rnd, inf = sys.argv[1:4]
sys.path.insert(0, os.getcwd())
fp_file = os.path.join(inf, "source_fingerprint.txt")      # written on the GPU box by tools/gpu_round.sh: the code that was measured
fingerprint = open(fp_file).read().strip() if os.path.exists(fp_file) else None
fetch = float(re.search(r"HBM fetch per step: ([0-9.]+) MB", open(p + "_pmc3_fetch.txt").read()).group(1))
write = float(re.search(r"HBM write per step: ([0-9.]+) MB", open(p + "_pmc4_write.txt").read()).group(1))
json.dump({"hbm_fetch_mb_per_step": fetch, "hbm_write_mb_per_step": write, "hbm_bytes_per_step": int((fetch + write) * 1e6),
           "frames_per_step": 128, "source_fingerprint": fingerprint, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB, FETCH x2 "
           "(gfx950 correction, MI355X_MICROARCH.md), summed over the dispatches of one bench step: " + p + "_pmc3_fetch.txt, "
           + p + "_pmc4_write.txt"}, open("profiles/%s/traffic.json" % rnd, "w"), indent=1)
print(open("profiles/%s/traffic.json" % rnd).read())
PY
fi
if [ -d "$TR" ]; then
  cp $TR/train_bench.log ${P}_train_bench.log
  cp $TR/train_nodes.log ${P}_train_nodes.log
  [ -f $TR/pytest_gpu.log ] && cp $TR/pytest_gpu.log ${P}_train_pytest_gpu.log
  for c in 3 4 5; do [ -f $TR/prof_cfg$c/stats_kernel_stats.csv ] && cp $TR/prof_cfg$c/stats_kernel_stats.csv ${P}_train_cfg${c}_kernel_stats.csv; done
fi
if [ -n "$4" ] && [ -d "gpurun_out/$4" ]; then   # bf16 training round
  B=gpurun_out/$4
  cp $B/train_bench.log ${P}_train_bf16_bench.log
  cp $B/train_nodes.log ${P}_train_bf16_nodes.log
  [ -f $B/prof_cfg4/stats_kernel_stats.csv ] && cp $B/prof_cfg4/stats_kernel_stats.csv ${P}_train_bf16_cfg4_kernel_stats.csv
fi
ls profiles/$ROUND
