cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_phaseshift_gpu.py -x -q -s > $O/tests1.txt 2>&1; echo "pytest rc $?" >> $O/tests1.txt
timeout 1800 python -m pytest tests/test_kirchhoff_gpu.py -x -q -k "ties" > $O/tests2.txt 2>&1; echo "pytest rc $?" >> $O/tests2.txt
echo "mfma: $(timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "mfma 4096: $(timeout 600 python3 profiles/tools/ps_quick.py 4096 2 2>&1 | tail -1)" >> $O/ps_quick.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/ps_stats/run -o x --output-format csv -- python3 $R/profiles/tools/ps_quick.py 8192 1 > $R/$O/ps_stats.log 2>&1
cp $(find $R/$O/ps_stats -name '*kernel_stats.csv' | head -1) $R/$O/ps_kernel_stats.csv
rocprofv3 --list-avail > $R/$O/list_avail.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/$O/ps_sq/run -o x --output-format csv -- python3 $R/profiles/tools/ps_quick.py 8192 1 > $R/$O/ps_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS -d $R/$O/ps_sq2/run -o x --output-format csv -- python3 $R/profiles/tools/ps_quick.py 8192 1 > $R/$O/ps_sq2.log 2>&1
cd $R
python3 - <<'PY' > $O/ps_pmc.txt 2>&1
import csv, glob, collections
for sub in ('ps_sq', 'ps_sq2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/r03d/%s/run/**/*counter_collection.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        if 'ps_' in k:
            print(k, {c: '%.4g' % (sum(v) / len(v)) for c, v in d.items()}, 'launches', max(len(v) for v in d.values()))
PY
timeout 1500 python3 profiles/tools/exchange_overlap.py c4r8 c3r8 c3r4 c3r2 --reserve 0,32 > $O/exchange_overlap.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace -d $R/$O/trace_r32/run -o x --output-format csv -- python3 $R/profiles/tools/exchange_overlap.py c4r8 c3r8 --reserve 32 --trace > $R/$O/trace_r32.log 2>&1
python3 $R/profiles/tools/trace_overlap.py $(find $R/$O/trace_r32 -name '*kernel_trace.csv' | head -1) 60 > $R/$O/trace_overlap_r32.txt 2>&1
cd $R
grep -E "passed|failed|Error" $O/tests1.txt | tail -5; tail -n 3 $O/tests2.txt; cat $O/ps_quick.txt; head -30 $O/ps_kernel_stats.csv | cut -c1-150; cat $O/ps_pmc.txt | cut -c1-600; tail -12 $O/exchange_overlap.txt; head -12 $O/trace_overlap_r32.txt
