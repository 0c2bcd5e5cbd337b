"""Where the first call of a fresh process goes: `impproc migrate` is one process per call, so the
first call IS the call.  Runs `bench.py --first-call stolt|phsh|kirch` children (fresh processes; this parent never
touches the GPU) with IMPDAR_TRACE=1, three ways:
  cold        HOME and ROCFFT_RTC_CACHE_PATH in an empty temp dir (no user kernel cache)
  warm-cache  once more against the cache the first child left
  default     the environment as it is
and prints every child's JSON line and its trace lines.   usage: first_call_probe.py [kinds...]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH = os.path.join(ROOT, 'bench.py')


def child(kind, env, tag):
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, '--first-call', kind], capture_output=True, text=True, timeout=600, env=env)
    wall = time.time() - t0
    line = [l for l in r.stdout.splitlines() if l.startswith('{"kind"')]
    rec = json.loads(line[-1]) if line else {'error': (r.stderr or r.stdout)[-400:]}
    rec['process_wall_s'] = round(wall, 2)
    print('== %s %s: %s' % (tag, kind, json.dumps(rec)), flush=True)
    for l in r.stderr.splitlines():
        if l.startswith('[impdar +') or 'rtc' in l.lower():
            print('   ' + l)
    sys.stdout.flush()
    return rec


def main():
    kinds = sys.argv[1:] or ['stolt', 'phsh', 'kirch']
    base = dict(os.environ, IMPDAR_TRACE='1')
    for kind in kinds:
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(base, HOME=tmp, XDG_CACHE_HOME=os.path.join(tmp, 'xdg'), ROCFFT_RTC_CACHE_PATH=os.path.join(tmp, 'rocfft_user_cache.db'),
                       ROCFFT_LOG_RTC_PATH=os.path.join(tmp, 'rtc.log'))
            child(kind, env, 'cold')
            for f in ('rocfft_user_cache.db', 'rtc.log'):
                p = os.path.join(tmp, f)
                print('   %s: %s bytes' % (f, os.path.getsize(p) if os.path.exists(p) else 'absent'))
            p = os.path.join(tmp, 'rtc.log')
            if os.path.exists(p):
                import re
                names = re.findall(r'(?:__global__[^\n]*?void\s+)(\w+)', open(p, errors='replace').read())
                print('   kernels rocFFT compiled at run time: %d %s' % (len(names), sorted(set(names))[:24]))
            env.pop('ROCFFT_LOG_RTC_PATH')
            child(kind, env, 'warm-cache')
        child(kind, base, 'default')


if __name__ == '__main__':
    main()
