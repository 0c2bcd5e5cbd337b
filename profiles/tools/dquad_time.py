"""ms per step of the float64 Kirchhoff ring kernel (kirch_dquad_kernel) at config 3 (10000 x 4096, resident), from bench.py's own
timed loop.   usage: dquad_time.py   (IMPDAR_HIP_LIB picks the build: profiles/tools/variant.sh kirchhoff.hip ...)"""
import subprocess, sys, json, os
out = subprocess.run([sys.executable, 'bench.py', '--dtype', 'f64', '--mode', 'exact', '--steps', '8', '--warmup', '2', '--no-cpu', '--no-pmc',
                      '--no-paths', '--no-e2e'], capture_output=True, text=True, env=dict(os.environ)).stdout
for l in out.splitlines():
    if l.startswith('{"metric"'):
        b = json.loads(l)
        print(json.dumps({'ms_per_step': round(b['ms_per_step'], 3), 'kernel': b['roofline'].get('kernel'), 'kernel_ms': b['roofline'].get('kernel_ms')}))
