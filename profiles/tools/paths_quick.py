"""Device ms of the config-2 (Stolt) and config-5 (Gazdag v(z)) paths, resident, a few repetitions (same-box A/B of
environment knobs: run it under different IMPDAR_* settings)."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench
r = bench.path_records(True, True)
print(json.dumps({k: {'device_ms': v.get('device_ms', v.get('ms_per_step')), 'kernel_ms': v.get('kernel_ms'), 'frac': (v.get('roofline') or {}).get('frac'), 'e2e': (v.get('end_to_end') or {}).get('host_call_ms')} for k, v in r.items()}))
