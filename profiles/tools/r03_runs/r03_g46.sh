cd $GRAFT_REPO_ROOT
echo "== default limits"; python3 profiles/tools/ps_layers.py 8192 2>&1 | grep "^{" | tail -5
echo "== LONG=24 PAD=4 SHORT=400"; IMPDAR_PS_MFMA_LONG=24 IMPDAR_PS_MFMA_PAD=4 IMPDAR_PS_MFMA_SHORT=400 python3 profiles/tools/ps_layers.py 8192 2>&1 | grep "^{" | tail -5
echo "== LONG=16 PAD=3 SHORT=200"; IMPDAR_PS_MFMA_LONG=16 IMPDAR_PS_MFMA_PAD=3 IMPDAR_PS_MFMA_SHORT=200 python3 profiles/tools/ps_layers.py 8192 2>&1 | grep "^{" | tail -5
timeout 600 python3 tests/tools/fuzz_phaseshift.py 120 71 2>&1 | tail -n 1
IMPDAR_PS_MFMA_LONG=24 IMPDAR_PS_MFMA_PAD=4 IMPDAR_PS_MFMA_SHORT=400 timeout 600 python3 tests/tools/fuzz_phaseshift.py 120 72 2>&1 | tail -n 1
