cd $GRAFT_REPO_ROOT
python -c "
import binius_amd, os
print('node', binius_amd.device_numa_node(0), 'affinity before', len(os.sched_getaffinity(0)))
print(binius_amd.bind_host_thread_to_device(0), len(os.sched_getaffinity(0)))
"
for i in 1 2 3 4; do
 echo "bound  $(python tools/small_rounds.py 2>/dev/null | tail -1 | cut -c1-150)"
 echo "free   $(BN_BIND_NUMA=0 python tools/small_rounds.py 2>/dev/null | tail -1 | cut -c1-150)"
done
for i in 1 2 3; do python bench.py --n-vars 24 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n24', d['ms_per_step'], d['config']['host_affinity'])"; BN_BIND_NUMA=0 python bench.py --n-vars 24 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n24 free', d['ms_per_step'])"; done
