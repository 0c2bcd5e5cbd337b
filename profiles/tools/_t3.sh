timeout 300 python3 profiles/tools/ps_quick64.py 8192 3 | cut -c1-500
timeout 300 python3 profiles/tools/ps_quick.py 8192 5 | cut -c1-500
timeout 1500 python -m pytest tests/test_phaseshift_gpu.py -x -q -k "vs_oracle_sizes or first_order or hermitian or golden or matrix_core_path_against or larger_size or transform_path" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
