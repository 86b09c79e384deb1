#!/bin/bash
# bench.py's sampling legs (faithful, hoisted, hoisted_graph) by ADX_RESNET_STREAMS, same box, alternating
for rnd in 1 2 3; do
  for n in 1 2; do
    echo -n "$rnd streams=$n "
    ADX_RESNET_STREAMS=$n python bench.py --no-cpu-baseline --no-roofline --no-train --no-deployed 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('faithful', d['value'], d['ms_per_step'], 'hoisted', d['hoisted']['value'], 'graph', d['hoisted_graph']['value'])"
  done
done
