cd $GRAFT_REPO_ROOT
BNH_PROF=1 python tools/small_rounds.py 2>&1 | grep "n_vars 12" | tail -3
BNH_PROF=1 python tools/small_rounds.py 2>&1 | grep "n_vars 8:" | tail -2
python tools/small_rounds.py 2>&1 | tail -3
