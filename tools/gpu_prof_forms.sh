#!/bin/bash
# usage: gpu_prof_forms.sh "ipb:form[,form...]" ...   rocprofv3 kernel stats of tools/ab_forms.py per (instances per box, form):
# one profiled process each, so that every kernel's average belongs to ONE form -> gpurun_out/prof_forms/<ipb>_<form>.txt
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_forms
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  ipb=${spec%%:*}; forms=${spec#*:}
  for form in ${forms//,/ }; do
    d=$R/gpurun_out/prof_forms/raw_${ipb}_${form}
    rm -rf $d
    timeout 240 rocprofv3 --kernel-trace --stats -d $d -o p --output-format csv -- python $R/tools/ab_forms.py --ipb $ipb --forms $form --reps 1 --steps 300 --sets 4 > $R/gpurun_out/prof_forms/${ipb}_${form}.log 2>&1
    f=$(find $d -name '*kernel_stats.csv' | head -1)
    python - "$f" "$ipb" "$form" <<'PY' | tee $R/gpurun_out/prof_forms/${ipb}_${form}.txt
import csv, sys
f, ipb, form = sys.argv[1:4]
rows = [r for r in csv.DictReader(open(f)) if 'bxi::' in r['Name']]
print(f'ipb {ipb} form {form}')
for r in rows:
    name = r['Name'].split('bxi::')[1].split('(')[0]
    print(f"  {name:32s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:7.2f}  max {float(r['MaxNs'])/1e3:7.2f}")
PY
    cp "$f" $R/gpurun_out/prof_forms/${ipb}_${form}_kernel_stats.csv 2>/dev/null
    rm -rf $d
  done
done
