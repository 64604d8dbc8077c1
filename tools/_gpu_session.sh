# Scratch driver for `gpurun -- 'bash tools/_gpu_session.sh'` (edited per session).  Last state: the full validation of a lease.
mkdir -p gpurun_out/validate
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/validate/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/validate/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/validate/bench.json 2> gpurun_out/validate/bench.err; echo "bench rc $?"
