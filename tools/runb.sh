#!/bin/bash
# usage: runb.sh workload steps
python3 bench.py --steps $2 --warmup 1 --no-cpu-baseline --no-extras --workload $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['config']['workload'], round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms | shadow', round(r['kernel_Mrays_per_s']), 'Mr/s', round(r['avg_launch_ms']*r['launches']/d['steps'],2), 'ms/frame | closest', round(r['other_kernels']['k_closest']['Mrays_per_s']), 'Mr/s', round(r['other_kernels']['k_closest']['avg_launch_ms']*r['other_kernels']['k_closest']['launches']/d['steps'],2), 'ms/frame | shade', round(r['other_kernels']['k_shade']['avg_launch_ms']*r['other_kernels']['k_shade']['launches']/d['steps'],2))
"
