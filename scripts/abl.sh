#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "" _abl1 _abl2 _abl3; do
  GDR_LIB_PATH=$PWD/generativedensification_amd/lib/libgdr_hip$v.so python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items() if 'render' in k})"
done
