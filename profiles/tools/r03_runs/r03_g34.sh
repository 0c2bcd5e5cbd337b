cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04e; mkdir -p $O; rm -rf $O/*
python3 profiles/tools/chain_stages.py > $O/chain.txt 2>&1; cat $O/chain.txt
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -- python3 profiles/tools/chain_stages.py > $O/run.txt 2>&1
python3 profiles/tools/timeline.py $O/tl > $O/timeline_chain.txt 2>&1
cat $O/timeline_chain.txt | head -70
find $O -name "*.csv" -size +4M -delete
