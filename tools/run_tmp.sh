cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ai
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r02ai/pytest.log
cat gpurun_out/r02ai/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02ai/smoke.log 2>&1
bash tools/collect_round.sh r02c > gpurun_out/r02ai/collect.log 2>&1
tail -3 gpurun_out/r02ai/smoke.log
