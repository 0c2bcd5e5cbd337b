cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04d; mkdir -p $O; rm -f $O/*.txt
timeout 1800 python -m pytest tests/test_phaseshift_gpu.py tests/test_phaseshift_sharded_gpu.py -x -q > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt
timeout 600 python3 profiles/tools/paths_quick.py >> $O/paths.txt 2>&1
timeout 600 python3 profiles/tools/ps_quick.py 8192 >> $O/paths.txt 2>&1
timeout 600 python3 profiles/tools/ps_quick.py 4096 >> $O/paths.txt 2>&1
cat $O/paths.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 profiles/tools/ps_quick.py 8192 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04d/st/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.reader(open(f)))[:12]:
    print(r[0][:70], r[1], r[3][:10])
PY
find $O -name "*.csv" -size +4M -delete
