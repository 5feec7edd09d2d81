cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
free -g > $O/mem.txt; nproc >> $O/mem.txt; cat /sys/fs/cgroup/cpu.max >> $O/mem.txt 2>&1
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
python bench.py > $O/bench_n28.json 2> $O/bench_n28.err
python bench.py --n-vars 24 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n24.json 2>/dev/null
python bench.py --n-vars 25 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n25.json 2>/dev/null
python tools/small_rounds.py > $O/small_rounds.jsonl 2>&1
BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_8ranks_one_gpu.json 2> $O/bench_8ranks_one_gpu.err
BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 8 --n-vars 20 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_8ranks_one_gpu_n20.json 2> $O/bench_8ranks_one_gpu_n20.err
tail -c 1200 $O/bench_n28.json; cat $O/small_rounds.jsonl
python -c "
import json
for n in (24,25,28):
    d=json.load(open('$O/bench_n%d.json'%n)); print(n, d['ms_per_step'], d['roofline']['frac'], d['kernels'])
for f in ('bench_8ranks_one_gpu','bench_8ranks_one_gpu_n20'):
    try:
        d=json.loads([l for l in open('$O/%s.json'%f) if l.startswith('{')][-1]); print(f, d['ms_per_step'], d['config']['sharding'][:60], d['alt_exchange'])
    except Exception as e: print(f, 'failed', e)
"
tail -5 $O/bench_8ranks_one_gpu.err
