"""Wall time of impdar_dev_alloc / impdar_dev_free (hipMalloc / hipFree) for radargram-sized device arrays."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import _hip
lib, ctx = _hip.load(), _hip.context()
for mb in (64, 164, 328, 512):
    ta, tf = [], []
    for i in range(5):
        t0 = time.perf_counter()
        d = _hip.DeviceArray(ctx, (mb << 20,), np.uint8)
        t1 = time.perf_counter()
        d.free()
        t2 = time.perf_counter()
        ta.append((t1 - t0) * 1e3); tf.append((t2 - t1) * 1e3)
    print('%4d MB: alloc %s ms, free %s ms' % (mb, ' '.join('%.2f' % x for x in ta), ' '.join('%.2f' % x for x in tf)), flush=True)
