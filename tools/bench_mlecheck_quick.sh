#!/bin/bash
# MLE-check prover timings: weighted prover vs the literal mirror (BN_MLECHECK=eager), n = 20, 24
for n in 20 24; do
  python tools/bench_mlecheck.py --n-vars $n 2>&1 | tail -3
  BN_MLECHECK=eager python tools/bench_mlecheck.py --n-vars $n 2>&1 | tail -3
done
