timeout 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/ab_forms.py --ipb 1 2 3 4 --forms auto,ready --reps 3 --steps 400 2>&1 | grep '^n'
for i in 1 2; do python bench.py --no-cpu-baseline --no-extras --no-kernel-timing 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', r['ms_per_step']*1e3)"; done
IPB=2 BXI_FLAGS=0 timeout 200 python tools/trace_forms.py 2>&1 | grep -v amdgpu.ids | tail -40
