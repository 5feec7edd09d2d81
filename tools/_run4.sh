cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; rm -rf $O; mkdir -p $O
tools/two_round_phases > $O/two_round_phases.txt 2>&1; grep -A10 "k_foldeval8<2>, n_in = 64 \|k_foldeval8<2>, n_in = 16384" $O/two_round_phases.txt
tools/small_round_phases 2>&1 | head -14
python tools/small_rounds.py | tee $O/small_rounds.jsonl
for W in 2 8; do
  BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2961$W bench.py --gpus $W --n-vars 13 --steps 50 --warmup 5 --no-cpu-baseline --no-prof > $O/bench_w${W}_n13.json 2> $O/bench_w${W}_n13.err
  python -c "
import json
try:
    d=json.loads([l for l in open('$O/bench_w${W}_n13.json') if l.startswith('{')][-1]); print('W=$W', round(d['ms_per_step'],4), [(a['exchange'][:12], a.get('exchange_us_per_round')) for a in d['alt_exchange']])
except Exception as e: print('W=$W failed', e)
"
done
timeout 1200 python -m pytest tests/test_gpu_two_round.py tests/test_gpu_sumcheck.py tests/test_gpu_sharded_vs_oracle.py tests/test_gpu_north_star.py -x -q 2>&1 | tail -3
for n in 20 24 25; do
    python bench.py --n-vars $n --steps 10 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=$n ms_per_step', d['ms_per_step'], d['verifier_check'], d['transcript_digest'])"
done
