#!/usr/bin/env bash
# A/B a diagnostics switch on ONE box: tools/ab_bench.sh RL_NO_SIDE_STREAM [rounds]
var=$1; rounds=${2:-2}
for i in $(seq $rounds); do
  for v in 1 0; do
    env $var=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], 'clouds/s', d['ms_per_step'], 'ms')"
  done
done
