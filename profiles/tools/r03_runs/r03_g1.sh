cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests/test_phaseshift_gpu.py -x -q > gpurun_out/r03a/ps_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r03a/ps_tests.txt
for cfg in "deep 1" "wide 1" "deep 0"; do set -- $cfg
  echo "shape=$1 herm=$2: $(IMPDAR_PS_SHAPE=$1 IMPDAR_PS_HERMITIAN=$2 timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> gpurun_out/r03a/ps_quick.txt
done
for cfg in "deep 1" "wide 1" "deep 0"; do set -- $cfg
  echo "4096 shape=$1 herm=$2: $(IMPDAR_PS_SHAPE=$1 IMPDAR_PS_HERMITIAN=$2 timeout 600 python3 profiles/tools/ps_quick.py 4096 2 2>&1 | tail -1)" >> gpurun_out/r03a/ps_quick.txt
done

for cfg in "deep 1" "wide 1" "deep 0"; do set -- $cfg
  echo "f64 shape=$1 herm=$2: $(IMPDAR_PS_SHAPE=$1 IMPDAR_PS_HERMITIAN=$2 timeout 600 python3 profiles/tools/ps_quick64.py 8192 2 2>&1 | tail -1)" >> gpurun_out/r03a/ps_quick.txt
done
tail -5 gpurun_out/r03a/ps_tests.txt; cat gpurun_out/r03a/ps_quick.txt
