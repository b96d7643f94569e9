#!/bin/bash
# A/B of two builds of libgdr_hip.so on one box: build/old/libgdr_hip.so against the in-tree one.
# bash scripts/gpu_ab_lib.sh "c4 c3 c2" [extra bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIB=generativedensification_amd/lib/libgdr_hip.so
cp $LIB build/new_libgdr_hip.so
WLS=${1:-"c4 c3 c2"}; shift
run() { timeout 300 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'])"; }
for wl in $WLS; do
  for rep in 1 2; do
    cp build/old/libgdr_hip.so $LIB; echo "== $wl old"; run --workload $wl "$@"; run --workload $wl --layout shell "$@"
    cp build/new_libgdr_hip.so $LIB; echo "== $wl new"; run --workload $wl "$@"; run --workload $wl --layout shell "$@"
  done
done
cp build/new_libgdr_hip.so $LIB
