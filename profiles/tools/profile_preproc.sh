cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s16pmc; mkdir -p $O; cd $R
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch/run -o x --output-format csv -- python3 profiles/tools/preproc_time.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write/run -o x --output-format csv -- python3 profiles/tools/preproc_time.py > $O/write.log 2>&1
python3 - <<'PY'
import csv, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/s16pmc'
for name in ('fetch','write'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(O+'/%s/run/x_counter_collection.csv'%name)):
        acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if any(t in k for t in ('ff_','fir_','lerp')): print(name, k, 'launches', len(v), 'mean KB', sum(v)/len(v))
PY
