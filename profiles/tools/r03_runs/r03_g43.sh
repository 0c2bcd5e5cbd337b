cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04m; rm -rf $O; mkdir -p $O
for v in ps_base ps_hotload ps_base ps_hotload; do
rocprofv3 --kernel-trace --output-format csv -d $O/st_$v -- python3 profiles/tools/ps_abl_run.py $PWD/build/diag/lib_$v.so > /dev/null 2>&1
python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
f = glob.glob('gpurun_out/r04m/st_%s/**/*kernel_trace.csv' % v, recursive=True)[0]
t = [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in csv.DictReader(open(f)) if 'ps_mfma_kernel' in r['Kernel_Name']]
print('%-14s ps_mfma_kernel ms: const %s   v(z) %s' % (v, ' '.join('%.2f' % x for x in t[:3]), ' '.join('%.2f' % x for x in t[3:])), flush=True)
PY
rm -rf $O/st_$v
done
