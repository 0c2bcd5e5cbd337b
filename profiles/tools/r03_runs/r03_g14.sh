cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
for cfg in "0 geo" "0 even" "1 geo" "1 even"; do set -- $cfg
echo "== pool=$1 pieces=$2" >> $O/e2e.txt
IMPDAR_HOST_POOL=$1 IMPDAR_DL_PIECES=$2 IMPDAR_TIMING=1 timeout 600 python3 profiles/tools/e2e_phases.py 2>&1 | grep -E "cached|wall" | head -8 >> $O/e2e.txt
done
cat $O/e2e.txt
