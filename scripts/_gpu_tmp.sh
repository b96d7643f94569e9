cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4h; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
b() { python bench.py "$@" --steps 10 --no-cpu-baseline --no-roofline --no-per-view-leg 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do for m in 0 1 2; do echo -n "c5 K7_VIEWS=$m rep$rep: "; GDR_K7_VIEWS=$m b --workload c5; done; done 2>&1 | tee $O/ab.txt
for m in 0 1; do echo -n "c5 shell K7_VIEWS=$m: "; GDR_K7_VIEWS=$m b --workload c5 --layout shell; done | tee -a $O/ab.txt
python bench.py --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('default', d['value'], 'dom', r['kernel'], r['frac'], r.get('frac_serial'), r.get('valu_issue_frac'), r.get('valu_issue_frac_serial'), 'path', r['path_frac'], r['path_frac_built'], r.get('path_frac_measured'))"
