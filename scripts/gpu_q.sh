#!/bin/bash
for e in "GDR_LAUNCH_HINTS=1" "GDR_LAUNCH_HINTS=0" "GDR_DEFER_D=0" "GDR_LAUNCH_HINTS=0 GDR_DEFER_D=0"; do echo "== $e"; env $e python scripts/absgrad_bench.py 2>/dev/null | cut -c1-120; done
