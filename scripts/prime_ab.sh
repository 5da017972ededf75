# sharded world-1 bench under the launcher, a few runs: are there step outliers left?
for i in 1 2 3 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$i bench.py --gpus 1 --steps 40 --mode sharded --no-cpu-baseline 2>gpurun_out/prime_err.txt > gpurun_out/prime_$i.json || { tail -5 gpurun_out/prime_err.txt; exit 1; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/prime_$i.json')); s=d['step_ms_rank0']; print('run $i', round(d['ms_per_step'],2), round(d['step_ms_median'],2), [(i,round(x,1)) for i,x in enumerate(s) if x>45])"
done
