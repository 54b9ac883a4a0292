#!/usr/bin/env python3
"""Turn the rocprofv3 outputs merged into gpurun_out/ into the tracked summaries under profiles/.

    python tools/summarize_profiles.py r02        (after `gpurun -- tools/gpu_profiles.sh r02`)

reads  gpurun_out/prof_eager/<round>_kernel_stats.csv                    (rocprofv3 --kernel-trace --stats)
       gpurun_out/pmc_FETCH_SIZE|pmc_WRITE_SIZE/<round>_counter_collection.csv  (one --pmc pass per counter)
       gpurun_out/prof_<row>/<round>_kernel_stats.csv, gpurun_out/<row>_bench.json, gpurun_out/block_trace.txt, bench_default.json
writes profiles/<round>_kernel_stats_eager.csv, profiles/<round>_hbm_traffic.json, profiles/<round>_summary.md
FETCH_SIZE is doubled for the streaming kernel, as MI355X_MICROARCH.md prescribes for 16 B/lane coalesced
reads on gfx950 (the counter tallies 128-B requests at 64 B); WRITE_SIZE is used as reported (KB).
"""
import collections
import csv
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else 'r02'
G, P = 'gpurun_out', 'profiles'
os.makedirs(P, exist_ok=True)
shutil.copy(f'{G}/prof_eager/{rnd}_kernel_stats.csv', f'{P}/{rnd}_kernel_stats_eager.csv')
for row in ('pairwise_op', 'dynamic_head', 'head_fused', 'discobox', 'levelset', 'tree_filter', 'two_launch', 'n64', 'n128', 'n128_fold', 'n128_long',
            'targets_n32', 'targets_n128'):
    if os.path.exists(f'{G}/prof_{row}/{rnd}_kernel_stats.csv'):
        shutil.copy(f'{G}/prof_{row}/{rnd}_kernel_stats.csv', f'{P}/{rnd}_{row}_kernel_stats.csv')
    if os.path.exists(f'{G}/{row}_bench.json'):
        shutil.copy(f'{G}/{row}_bench.json', f'{P}/{rnd}_{row}_bench.json')
for src, dst in (('block_trace.txt', f'{rnd}_block_trace.txt'), ('bench_default.json', f'{rnd}_bench_default.json'), ('pairwise_op_pmc.txt', f'{rnd}_pairwise_op_pmc.txt')):
    if os.path.exists(f'{G}/{src}'):
        shutil.copy(f'{G}/{src}', f'{P}/{dst}')
stats = {}
for r in csv.DictReader(open(f'{P}/{rnd}_kernel_stats_eager.csv')):
    if 'bxi::' in r['Name']:
        key = r['Name'].split('bxi::')[1].split('(')[0].split('<')[0]
        stats[key] = dict(calls=int(r['Calls']), avg_us=float(r['AverageNs']) / 1e3, min_us=float(r['MinNs']) / 1e3,
                          max_us=float(r['MaxNs']) / 1e3)
traffic = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    path = f'{G}/pmc_{c}/{rnd}_counter_collection.csv'
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if 'bxi::' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('bxi::')[1].split('(')[0].split('<')[0]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        v = v[len(v) // 4:]                       # drop warm-up launches
        traffic[k][c + '_KB'] = sum(v) / len(v)
out = {}
for k, t in traffic.items():
    fetch = t.get('FETCH_SIZE_KB', 0.0) * 1024 * 2      # gfx950: x2 for wide coalesced reads
    write = t.get('WRITE_SIZE_KB', 0.0) * 1024
    out[k] = dict(fetch_bytes_corrected=fetch, write_bytes=write, hbm_bytes=fetch + write, raw=t)
json.dump(dict(round=rnd, command='python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-extras', kernels=out,
               note='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950 correction, '
                    'MI355X_MICROARCH.md section HBM); per launch, mean over the timed launches'),
          open(f'{P}/{rnd}_hbm_traffic.json', 'w'), indent=1)
with open(f'{P}/{rnd}_summary.md', 'w') as f:
    f.write(f'# rocprofv3 summary, round {rnd[1:]} (MI355X, `rocprofv3 --kernel-trace --stats -- python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-kernel-timing --no-extras`)\n\n')
    f.write('| kernel | calls | avg us | min us | max us | HBM read (PMC, corrected) MB | HBM write (PMC) MB |\n|---|---:|---:|---:|---:|---:|---:|\n')
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]['avg_us']):
        t = out.get(k, {})
        f.write(f"| {k} | {s['calls']} | {s['avg_us']:.2f} | {s['min_us']:.2f} | {s['max_us']:.2f} | "
                f"{t.get('fetch_bytes_corrected', float('nan')) / 1e6:.2f} | {t.get('write_bytes', float('nan')) / 1e6:.2f} |\n")
print(open(f'{P}/{rnd}_summary.md').read())

# ---- the "next" rows: one table per tracked rocprofv3 kernel-stats file under profiles/ ------------------
BENCH = 'bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-kernel-timing --no-extras'
EXTRA = [('two_launch', 'the same evaluation in its two-launch form (`--flags 2` = BXI_EVAL_TWO_LAUNCHES)', BENCH + ' --flags 2'),
         ('n64', '64 instances (2 per box), default form', BENCH + ' --inst-per-box 2 --sets 6'),
         ('n128', '128 instances (4 per box: what configs/boxinst/boxinst_r50_fpn_1x_coco.py:65,125 trains at), default form', BENCH + ' --inst-per-box 4 --sets 6'),
         ('targets_n32', 'targets ahead at 32 instances: bxi_boxinst_targets_f32 (targets_pool + targets_pred) and the evaluation with BXI_EVAL_TARGETS_READY', 'tools/ab_forms.py --ipb 1 --forms ready,targets_only --reps 1 --steps 300 --sets 6'),
         ('targets_n128', 'targets ahead at 128 instances', 'tools/ab_forms.py --ipb 4 --forms ready,targets_only --reps 1 --steps 300 --sets 6'),
         ('dynamic_head', 'f-2 dynamic mask head', 'tools/bench_dynamic_head.py'),
         ('head_fused', 'f-2 head fused into the evaluation', 'tools/bench_head_fused.py'),
         ('discobox', 'f-3 DiscoBox MeanField / mil_loss / dice_loss', 'tools/bench_discobox.py'),
         ('levelset', 'f-4 BoxProjectionLoss / LevelsetLoss / LCM', 'tools/bench_levelset.py'),
         ('tree_filter', 'f-4 tree_filter (mst / bfs / refine)', 'tools/bench_tree_filter.py'),
         ('pairwise_op', 'a-9..a-11 op-level pairwise_nlog (f32 / f64)', 'tools/bench_pairwise_op.py')]
with open(f'{P}/{rnd}_summary.md', 'a') as f:
    for key, title, cmd in EXTRA:
        path = f'{P}/{rnd}_{key}_kernel_stats.csv'
        if not os.path.exists(path):
            continue
        f.write(f'\n## {title} (`rocprofv3 --kernel-trace --stats -- python {cmd}`, `{os.path.basename(path)}`)\n\n')
        f.write('| kernel | calls | avg us | min us | max us |\n|---|---:|---:|---:|---:|\n')
        rows = [r for r in csv.DictReader(open(path)) if 'bxi::' in r['Name']]
        for r in sorted(rows, key=lambda r: -float(r['AverageNs'])):
            name = r['Name'].split('bxi::')[1].split('(')[0]
            f.write(f"| {name} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} |\n")
print(open(f'{P}/{rnd}_summary.md').read()[-2500:])
