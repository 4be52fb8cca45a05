for round in 1 2; do
for v in gf_base gf_2plane; do
  cp tools/debug/_ab/$v.hip desco_amd/csrc/gossip_fused.hip
  make -C desco_amd/csrc > /dev/null 2>&1
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-x1 --no-attainable --no-train 2>/dev/null | V=$v python -c "
import sys, json, os
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']['gossip_fused_kernel']
        print(os.environ['V'], round(k['ms'] / k['calls'], 4), 'ms/launch; step', round(d['ms_per_step'], 2), 'value', round(d['value']))"
done
done
cp tools/debug/_ab/gf_2plane.hip desco_amd/csrc/gossip_fused.hip; make -C desco_amd/csrc > /dev/null 2>&1
python -m pytest tests/test_model_gpu.py -q -s -k "gossip" 2>&1 | grep -E "parity|passed|failed"
