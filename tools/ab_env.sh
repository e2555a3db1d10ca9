#!/bin/bash
# A/B of environment-selected schedules: tools/ab_env.sh <out> "<bench args>" "<ENV=..;ENV=..>" ["<ENV..>" ...]   (GPU box)
out=$1; shift; args=$1; shift
: > $out
for e in "$@"; do
  line=$(env $e python3 bench.py $args --no-cpu-baseline --no-wosac-shape --new-scenes 0 --profile-steps 0 --detail-file - 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %10.0f  ms/step %.4f' % (d['value'], d['ms_per_step']))")
  echo "$args | $e | $line" >> $out
done
cat $out
