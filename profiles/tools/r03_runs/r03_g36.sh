cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04g; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/st -- python3 profiles/tools/ps_quick.py 8192 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04g/st/**/*kernel_trace.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'ps_setup' in r['Kernel_Name'] or 'ps_mfma_kernel' in r['Kernel_Name']:
        print(r['Kernel_Name'][:30], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
PY
rm -rf $O/st
