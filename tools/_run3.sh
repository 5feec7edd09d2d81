cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; rm -rf $O; mkdir -p $O
tools/two_round_phases > $O/two_round_phases.txt 2>&1; cat $O/two_round_phases.txt
tools/small_round_phases 2>&1 | head -14
for W in 2 4 8; do
 for NV in 13 20; do
  BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2961$W bench.py --gpus $W --n-vars $NV --steps 50 --warmup 5 --no-cpu-baseline --no-prof > $O/bench_w${W}_n$NV.json 2> $O/bench_w${W}_n$NV.err
  python -c "
import json
try:
    d=json.loads([l for l in open('$O/bench_w${W}_n$NV.json') if l.startswith('{')][-1]); print('W=$W n=$NV', round(d['ms_per_step'],4), d['config']['sharding'][-80:], json.dumps(d['alt_exchange']))
except Exception as e: print('W=$W n=$NV failed', e)
"
 done
done
timeout 600 python -m pytest tests/test_gpu_multirank.py -x -q 2>&1 | tail -3
