#!/bin/bash
# per-kernel event timings at several particle counts (latency floor vs throughput)
for n in 65536 262144 1048576; do
python bench.py --particles $n --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-batch --no-pmmh 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('N=%d us/obs %.2f' % ($n, j['sweep']['us_per_observation']), {k.split('(')[0][:22]: round(v['avg_us'],2) for k,v in j['kernels'].items() if v['launches']>10})
"
done
