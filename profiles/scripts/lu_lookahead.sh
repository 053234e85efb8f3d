for la in 0 1; do echo "== LOOKAHEAD=$la"; NLH_LU_LOOKAHEAD=$la python profiles/lu_time.py 256 257 300 513 700 1000 1024 2>&1 | grep "n="; done
python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "lu" 2>&1 | tail -3
python -m pytest tests/test_gpu_solvers.py tests/test_gpu_configs.py -x -q -m gpu -k "newton or c3" 2>&1 | tail -3
python tests/soak_lu.py 2>&1 | tail -2
