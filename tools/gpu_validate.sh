#!/bin/bash
# full GPU suite + smoke + extended fuzz of the loss path + the two bench invocations (default, the driver's)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --maxfail=20 -p no:cacheprovider > gpurun_out/validate_pytest_full.log 2>&1
tail -6 gpurun_out/validate_pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python tools/extended_fuzz.py 24 500 loss > gpurun_out/validate_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/validate_fuzz.log
timeout 600 python bench.py > gpurun_out/validate_bench.json 2> gpurun_out/validate_bench.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/validate_bench_driver.json 2> gpurun_out/validate_bench_driver.err; echo "bench-driver rc=$?"
python - <<'PY'
import json
for f in ('gpurun_out/validate_bench.json', 'gpurun_out/validate_bench_driver.json'):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'us/step %.2f' % (r['ms_per_step'] * 1e3), 'frac', r['roofline']['frac'], 'sol_us', r['roofline'].get('sol_us'),
              {k: (round(v['us_per_step'], 2) if isinstance(v, dict) and 'us_per_step' in v else None) for k, v in r.get('extras', {}).items() if k.startswith('n')})
    except Exception as e:
        print(f, 'FAILED', e)
PY
