#!/bin/bash
# Same-box A/B of two builds of libhobbit_hip.so (e.g. -DHOBBIT_FMUL_U128 against the default): bench.py alternates between them in one gpurun call.
# usage: scripts/ab_lib.sh <other.so> [steps]
OTHER=$1; STEPS=${2:-20}
for rep in 1 2; do
  for lib in default $OTHER; do
    if [ "$lib" = default ]; then unset HOBBIT_HIP_LIB; else export HOBBIT_HIP_LIB=$(realpath $lib); fi
    python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
k = d.get('kernels_ms_extra_profiled_step', {})
top = sorted(k.items(), key=lambda kv: -kv[1])[:12]
print('$lib', 'ms_per_step %.3f' % d['ms_per_step'], ' '.join('%s %.2f' % (a, b) for a, b in top))
"
  done
done
