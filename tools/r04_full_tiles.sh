cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/full
for rep in 1 2; do
for b in old abl0 abl4 abl8 abl12; do
  echo "== $b (rep $rep)"
  timeout 60 tools/gram_bench_$b 18 2>&1 | grep -E "fused N=.*(fold|sums)|ALL OK|FAILED" | head -6
  for l in 24 26 27; do timeout 120 tools/gram_bench_$b $l prof 30 2>&1 | grep -E "fused N=.*sustained" | tail -2; done
done
done > gpurun_out/full/times.txt 2>&1
cat gpurun_out/full/times.txt
