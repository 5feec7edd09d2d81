cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_two_round.py -x -q > $O/pytest_two_round.log 2>&1; echo "rc=$?" >> $O/pytest_two_round.log
tail -15 $O/pytest_two_round.log
timeout 1200 python -m pytest tests/test_gpu_sumcheck.py tests/test_gpu_at_size.py tests/test_gpu_sharded_vs_oracle.py tests/test_gpu_multirank.py tests/test_gpu_lazy_vs_eager.py tests/test_gpu_layer.py tests/test_gpu_cpp_conformance.py -x -q > $O/pytest_rest.log 2>&1; echo "rc=$?" >> $O/pytest_rest.log
tail -8 $O/pytest_rest.log
for L in 16 17 18; do
  echo "== BN_TWO_ROUND_MAX_LOG2=$L"
  BN_TWO_ROUND_MAX_LOG2=$L python tools/small_rounds.py 2>&1 | tee $O/small_rounds_$L.jsonl
  for n in 20 24 25; do
    BN_TWO_ROUND_MAX_LOG2=$L python bench.py --n-vars $n --steps 10 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=$n L=$L ms_per_step', d['ms_per_step'], d['verifier_check'], d['transcript_digest'])"
  done
done
echo "== BN_TWO_ROUND=0"
BN_TWO_ROUND=0 python tools/small_rounds.py 2>&1 | tee $O/small_rounds_off.jsonl
for n in 20 24 25; do
  BN_TWO_ROUND=0 python bench.py --n-vars $n --steps 10 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=$n off ms_per_step', d['ms_per_step'], d['verifier_check'], d['transcript_digest'])"
done
python bench.py --n-vars 24 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n24.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_n24.json')); print(d['ms_per_step'], d['kernels'])"
timeout 1200 python -m pytest tests/test_gpu_north_star.py -x -q --durations=5 > $O/pytest_north_star.log 2>&1; echo "rc=$?" >> $O/pytest_north_star.log
tail -12 $O/pytest_north_star.log
BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_8ranks_one_gpu.json 2> $O/bench_8ranks_one_gpu.err
python -c "
import json
for f in ('bench_8ranks_one_gpu',):
    try:
        d=json.loads([l for l in open('$O/%s.json'%f) if l.startswith('{')][-1]); print(f, d['ms_per_step'], d['config']['sharding'][:60], d['alt_exchange'])
    except Exception as e: print(f, 'failed', e)
"
