cd $GRAFT_REPO_ROOT
tools/trace_cmd.sh r3f/trace_fold python tools/run_fold_only.py; tail -4 gpurun_out/r3f/trace_fold/per_launch.jsonl
tools/pmc_cmd.sh r3f/pmc_fold python tools/run_fold_only.py 2>&1 | tail -12
