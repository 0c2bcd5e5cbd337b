"""Samples the GPU's clock / busy / power and the processes holding a KFD context every ~20 ms (sysfs, no rocm-smi) until killed.
usage: gpu_sampler.py OUT    (one line per sample: unix time, sclk MHz, busy %, power W, kfd pids)"""
import glob, os, sys, time
out = open(sys.argv[1], 'w')
cards = [c for c in glob.glob('/sys/class/drm/card*/device') if os.path.exists(c + '/pp_dpm_sclk')]
def rd(p):
    try:
        return open(p).read()
    except Exception:
        return ''
while True:
    row = ['%.3f' % time.time()]
    for c in cards:
        sclk = [l for l in rd(c + '/pp_dpm_sclk').splitlines() if l.strip().endswith('*')]
        busy = rd(c + '/gpu_busy_percent').strip()
        pw = ''
        for h in glob.glob(c + '/hwmon/hwmon*/power1_average') + glob.glob(c + '/hwmon/hwmon*/power1_input'):
            pw = rd(h).strip()
            break
        row.append('%s|%s|%s' % (sclk[0].split(':')[1].strip() if sclk else '?', busy, pw))
    try:
        row.append(','.join(sorted(os.listdir('/sys/class/kfd/kfd/proc'))))
    except Exception:
        row.append('?')
    out.write(' '.join(row) + '\n')
    out.flush()
    time.sleep(0.02)
