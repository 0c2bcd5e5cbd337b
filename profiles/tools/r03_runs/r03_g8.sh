cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
R=$GRAFT_REPO_ROOT
cat > /tmp/kq.py <<'PY'
import sys, os, json, io, contextlib
import ctypes as C
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
lib, ctx = _hip.load(), _hip.context()
n = 8192
rng = np.random.default_rng(0)
geo = synth.geometry(n, n)
x = rng.standard_normal((n, n)).astype(np.float32)
ms = []
for i in range(3):
    d = RadarData(None)
    d.data, (d.snum, d.tnum) = x, x.shape
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    d.to_device()
    with contextlib.redirect_stdout(io.StringIO()):
        d.migrate('phsh', vel=1.69e8, htaper=100, vtaper=1000)
    v = C.c_float()
    _hip.check(lib.impdar_ctx_last_kernel_ms(ctx, C.byref(v)), 'k')
    ms.append(round(v.value, 3))
    d._dev.free(); d._dev = None
print('const-v kernel_ms', ms)
PY
for rep in 1 2; do
echo "default: $(timeout 600 python3 /tmp/kq.py 2>&1 | tail -1)" >> $O/ablation.txt
for v in nomfma nogen nobgen nogenb; do
echo "$v: $(IMPDAR_HIP_LIB=$R/build/diag/lib_$v.so timeout 600 python3 /tmp/kq.py 2>&1 | tail -1)" >> $O/ablation.txt
done
done
cat $O/ablation.txt
