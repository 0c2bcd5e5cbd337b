#!/bin/bash
# kernel + copy timeline of the one-shot Kirchhoff call (GPU box): profiles/tools/oneshot_trace.sh <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
E2E_CALLS=3 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/run -o x --output-format csv -- python3 $R/profiles/tools/e2e_f32.py > $O/log.txt 2>&1 </dev/null
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
ev = []
for f in glob.glob(o + '/run/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:40] + ' grid=' + r.get('Grid_Size_X', r.get('Grid_Size', '?')) + ' q=' + r.get('Queue_Id', '?')))
for f in glob.glob(o + '/run/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', r.get('Name', '?'))))
ev.sort()
# the last call: events after the last gap of > 30 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[:i][-50:]) > 30e6: cut = i
t0 = ev[cut][0]
with open(o + '/timeline.txt', 'w') as f:
    for a, b, n in ev[cut:]:
        f.write('%9.3f %9.3f %8.3f  %s\n' % ((a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, n))
print(open(o + '/timeline.txt').read())
PY
