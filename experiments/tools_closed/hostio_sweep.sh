for e in 1 2 3; do for k in 20 200 2000; do python bench.py --engines $e --steps $k --warmup 5 --no-others --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); h=r['pcie_inclusive']
print('engines',r['config']['engines_per_gpu'],'steps',r['steps'],'resident',r['value'],'host',h['value'],h['of_resident'],'kms',h['kernel_avg_launch_ms'])"; done; done
