#!/bin/bash
# rebuild the library from the repo root whatever the caller's directory is (hipcc only for what changed)
cd "$(dirname "$0")/../.." && python3 -c "from impdar_amd import build; build.build()" 2>&1 | grep -E "error|warning: unused|hipcc.*-shared" | cut -c1-100
