for i in 1 2; do for p in 3 4 5 6 8; do
  timeout 200 python bench.py --no-cpu-baseline --no-train-configs --steps 20 --warmup 3 --sustained-seconds 1.5 --pipeline $p 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipeline=$p value %.0f sustained %.0f' % (d['value'], d['windows']['sustained_value']))"
done; done
