cd $GRAFT_REPO_ROOT
tools/final_measure.sh
ls gpurun_out/final | head -50
python -c "
import json
d=json.load(open('gpurun_out/final/bench_pmc.json')); print(d.get('build'), d.get('csrc_sha16'), list(d['workloads'].keys()))
for w,v in d['workloads'].items():
    for k,e in v.items():
        if 'foldeval_mfma' in k or 'fp4' in k or 'foldeval8' in k: print(w,k,e)
"
