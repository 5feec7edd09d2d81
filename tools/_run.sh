cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
free -g > $O/mem.txt; nproc >> $O/mem.txt; cat /sys/fs/cgroup/cpu.max >> $O/mem.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -k "not peer" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py > $O/bench_n28.json 2> $O/bench_n28.err
python bench.py --n-vars 24 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n24.json 2>/dev/null
python bench.py --n-vars 25 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n25.json 2>/dev/null
python tools/small_rounds.py > $O/small_rounds.jsonl 2>&1
tail -c 1500 $O/bench_n28.json; cat $O/small_rounds.jsonl
python -c "
import json
for n in (24,25,28):
    d=json.load(open('$O/bench_n%d.json'%n)); print(n, d['ms_per_step'], d['roofline']['frac'], d['kernels'])
"
