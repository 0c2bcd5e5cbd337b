#!/bin/bash
# Effective clock of kirch_quad_kernel under different builds: GRBM_GUI_ACTIVE (sum over the 8 XCDs) / 8 / kernel time,
# LDS-busy share = SQ_LDS_IDX_ACTIVE / 256 CUs / cycles.  usage: clock_probe.sh <lib name under build/diag | base> ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in "$@"; do
  if [ "$n" = base ]; then L=$R/impdar_amd/csrc/libimpdar_hip.so; else L=$R/build/diag/lib_$n.so; fi
  O=$R/gpurun_out/clock_$n
  rm -rf $O; mkdir -p $O
  IMPDAR_HIP_LIB=$L rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/run -o x --output-format csv -- python3 $R/bench.py --pmc-child --steps 6 --warmup 2 --data-child synthetic > $O/log 2>&1
  python3 - "$O" "$n" <<'PY'
import csv, glob, sys, collections
o, name = sys.argv[1], sys.argv[2]
c = collections.defaultdict(list)
for f in glob.glob(o + '/run/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'kirch_quad_kernel' in r['Kernel_Name']:
            c[r['Counter_Name']].append(float(r['Counter_Value']))
d = []
for f in glob.glob(o + '/run/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'kirch_quad_kernel' in r['Kernel_Name']:
            d.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
if not d or not c:
    print(name, 'no data'); sys.exit()
ms = sorted(d)[len(d) // 2]
gui = sorted(c['GRBM_GUI_ACTIVE'])[len(c['GRBM_GUI_ACTIVE']) // 2]
lds = sorted(c['SQ_LDS_IDX_ACTIVE'])[len(c['SQ_LDS_IDX_ACTIVE']) // 2]
cyc = gui / 8
print('%-10s kernel %.3f ms  clock %.3f GHz  cycles/XCD %.3e  LDS busy %.1f %% of CU cycles' % (name, ms, cyc / ms / 1e6, cyc, 100 * lds / 256 / cyc))
PY
done
