#!/usr/bin/env python3
"""Developer helper: summarise gpurun_out/ (rocprof kernel stats, block trace, bench lines)."""
import csv, json, numpy as np, os
g = 'gpurun_out'
for r in csv.DictReader(open(f'{g}/prof_eager/r01_kernel_stats.csv')):
    if 'bxi' in r['Name']: print('%-36s calls %s avg %.2f us min %.2f max %.2f' % (r['Name'].replace('void ','')[:36], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
for mode in ('eager', 'graph'):
    try:
        r = json.loads(open(f'{g}/bench_{mode}.json').read().strip().splitlines()[-1])
        print(mode, 'value %.0f img/s  us/step %.2f' % (r['value'], r['ms_per_step']*1e3), {k: round(v['avg_us'], 2) for k, v in r.get('kernels', {}).items()})
    except Exception as e: print(mode, 'n/a', e)
t = np.load(f'{g}/trace.npz')['trace'].astype(np.float64)
us = lambda x: x * 0.01
t0 = t[0][t[0]>0].min()
s1 = t[0]; m = s1[:,0]>0; ns = 800; nb = int(m.sum())
ntab = 32; pp = s1[ntab:nb-ns]; sp = s1[nb-ns:nb]; print('  table waves dur', np.round(us(s1[:ntab,1]-s1[:ntab,0]),2).tolist()[:6], 'max', us(s1[:ntab,1]-s1[:ntab,0]).max())
print('stage1: blocks', nb, 'last start %.2f last end %.2f' % (us(s1[m,0].max()-t0), us(s1[m,1].max()-t0)))
print('  stream waves: dur mean %.2f max %.2f | issue loads+zero-fill %.2f | tables (t==0 only) %.2f | wait data+column max %.2f | row butterflies+store %.2f' % ((us(sp[:,1]-sp[:,0]).mean(), us(sp[:,1]-sp[:,0]).max()) + tuple(us(sp[:,b_]-sp[:,a_]).mean() for a_,b_ in [(0,2),(2,3),(3,4),(4,1)])))
print('  pool waves  : dur mean %.2f max %.2f | loads issued+data arrived %.2f | barrier %.2f | de-normalise+Lab+store %.2f' % ((us(pp[:,1]-pp[:,0]).mean(), us(pp[:,1]-pp[:,0]).max()) + tuple(us(pp[:,b_]-pp[:,a_]).mean() for a_,b_ in [(0,2),(2,3),(3,1)])))
print('  end-time quantiles: stream', np.quantile(us(sp[:,1]-t0),[.5,.9,1]).round(2), 'pool', np.quantile(us(pp[:,1]-t0),[.5,.9,1]).round(2))
b = t[1]; m = b[:,0]>0
if m.any():
    bs = b[m,0].min()
    print('box: tile blocks', m.sum(), 'first start %.2f (stage1 end +%.2f) last start +%.2f' % (us(bs-t0), us(bs-s1[s1[:,1]>0,1].max()), us(b[m,0].max()-bs)))
    hit = m & (b[:,2]>0); nh = m & ~hit
    print('  hit blocks', hit.sum())
    for a_,b_,nm in [(0,1,'decide'),(1,2,'loads+stage'),(2,4,'affinity+pairwise'),(4,5,'partials+store')]:
        d = us(b[hit,b_]-b[hit,a_]); print('  hit phase %-14s mean %.2f med %.2f max %.2f' % (nm, d.mean(), np.median(d), d.max()))
    print('  hit last end +%.2f ; hit block total mean %.2f max %.2f' % (us(b[hit,5].max()-bs), us(b[hit,5]-b[hit,0]).mean(), us(b[hit,5]-b[hit,0]).max()), ' start quantiles', np.quantile(us(b[m,0]-bs),[0,.25,.5,.75,1]).round(2))

    d = us(b[hit,6]-b[hit,5]); print('  hit phase arrive          mean %.2f med %.2f max %.2f ; last tile end +%.2f' % (d.mean(), np.median(d), d.max(), us(b[hit,6].max()-bs)))
ld = t[2]; ml = ld[:,0]>0
if ml.any():
    print('leaders: n', ml.sum(), 'start +%.2f' % us(ld[ml,0].min()-bs), '| loads+maxima %.2f | sums %.2f | grads+state %.2f | arrive %.2f | last end +%.2f' % tuple([us(ld[ml,b_]-ld[ml,a_]).mean() for a_,b_ in [(0,1),(1,2),(2,3),(3,4)]] + [us(ld[ml,4].max()-bs)]))
