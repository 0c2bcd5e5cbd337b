cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
IMPDAR_PS_STAMPS=1 timeout 600 python3 profiles/tools/ps_quick.py 8192 1 > $O/stamps.txt 2>&1
grep "stamps round" $O/stamps.txt | head -40
