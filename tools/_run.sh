cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/pytest_gpu.log
bash tools/final_measure.sh > gpurun_out/final_measure.log 2>&1
cat gpurun_out/pytest_gpu.log
