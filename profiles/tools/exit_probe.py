import subprocess, sys, os, tempfile, time
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
for kind in ('stolt', 'phsh'):
    for i in range(4):
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, HOME=tmp, XDG_CACHE_HOME=tmp + '/x', ROCFFT_RTC_CACHE_PATH=tmp + '/c.db')
            t0 = time.time()
            r = subprocess.run([sys.executable, R + '/bench.py', '--first-call', kind], capture_output=True, text=True, env=env)
            print(kind, i, 'rc', r.returncode, 'wall %.2f' % (time.time() - t0), 'stderr tail:', r.stderr[-200:].replace('\n', ' | '))
