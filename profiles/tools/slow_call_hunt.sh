R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python profiles/tools/gpu_sampler.py gpurun_out/r05_hunt_samples.txt & SP=$!
for i in 1 2 3; do
  IMPDAR_TRACE=1 python bench.py --no-e2e > gpurun_out/r05_hunt_$i.json 2> gpurun_out/r05_hunt_$i.err
  python - $i <<'PY'
import re, sys
i = sys.argv[1]
idle = None
for l in open('gpurun_out/r05_hunt_%s.err' % i):
    m = re.search(r'unix ([0-9.]+)\] ps_runs: (stream idle|ps_runs_kernel done)', l)
    if not m: continue
    t = float(m.group(1))
    if m.group(2) == 'stream idle': idle = t
    elif idle is not None:
        print('run', i, 'ps_runs_kernel %.1f ms at unix %.3f' % ((t - idle) * 1e3, idle))
        if (t - idle) > 0.03:
            rows = [r for r in open('gpurun_out/r05_hunt_samples.txt') if idle - 0.15 < float(r.split()[0]) < t + 0.1]
            print(''.join(rows[:40]))
PY
done
kill $SP
