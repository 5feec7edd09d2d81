# one prover, k disjoint claims, about the same total data (m 2^n ~ 2^28.6 elements) dealt to more and smaller arrays
# (BENCH_PAD_ELEMS=p: p elements left free between consecutive arrays -- arrays NOT at power-of-two strides; no effect measured)
for kn in "3 26" "6 25" "12 24" "25 23" "50 22" "100 21" "5 24" "7 24" "9 24"; do set -- $kn
python tools/bench_piop.py claims --n-vars $2 --k $1 --group 1 --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k', d['k'], 'n', d['n_vars'], 'ms', d['ms_per_prove'], 'frac', d['whole_prove_frac_of_64mN'], 'group_launch_frac', d.get('group_launch_frac'), d['prof_ms'])"
done
for kk in "100 keccak 22" "100 keccak 18" "40 star 22" "50 piop 22" "64 disjoint 22"; do set -- $kk
python tools/bench_piop.py claims --n-vars $3 --k $1 --kind $2 --group 1 --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kind', d['kind'], 'k', d['k'], 'm', d['m'], 'n', d['n_vars'], 'ms', d['ms_per_prove'], 'frac', d['whole_prove_frac_of_64mN'], d['prof_ms'], 'verifier', d['verifier_check'])"
done
