#!/usr/bin/env python3
"""What a training loop sees: GPU time the loss evaluation adds ONCE PER ITERATION behind other kernels (cold instruction / scalar caches, a drained
queue), per form of the evaluation.  Stand-in for the rest of the iteration: two 4096^3 bf16 matrix products (or, --standin copy, a 256 MB copy).

    python tools/cold_cost.py [--ipb 1 4] [--standin mm|copy|none]
"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ipb', type=int, nargs='+', default=[1, 4])
    ap.add_argument('--standin', default='mm')
    ap.add_argument('--iters', type=int, default=200)
    args = ap.parse_args()
    import __graft_entry__ as entry
    entry.build()
    import bench
    from boxinstseg_amd import _lib, functional as Fh, synthetic
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    ones = torch.ones(2, device=dev)
    stream = torch.cuda.Stream(device=dev)
    st = stream.cuda_stream
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16); b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16); c = torch.empty_like(a)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big2 = torch.empty_like(big)

    def standin():
        if args.standin == 'mm':
            torch.mm(a, b, out=c); torch.mm(c, b, out=a)
        elif args.standin == 'copy':
            big2.copy_(big)

    L = _lib
    forms = {'auto': 0, 'two_launches': L.EVAL_TWO_LAUNCHES, 'no_stay_on': L.EVAL_SHARED_DEVICE}
    out = {}
    for ipb in args.ipb:
        sets = [bench.EvalSet(lib, Fh, synthetic, dev, seed=7000 + i, inst_per_box=ipb, ones=ones, flags=0) for i in range(6)]
        res = {}
        with torch.cuda.stream(stream):
            def timed(fn):
                for i in range(20): fn(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for i in range(args.iters): fn(i)
                e1.record(stream); torch.cuda.synchronize()
                return e0.elapsed_time(e1) / args.iters * 1e3
            base = timed(lambda i: standin())
            for name, form in forms.items():
                def it(i):
                    standin()
                    rc = lib.bxi_boxinst_eval_f32(*sets[i % 6].eval_args[:-1], C.c_uint(form), st)
                    assert rc == 0
                def b2b(i):
                    rc = lib.bxi_boxinst_eval_f32(*sets[i % 6].eval_args[:-1], C.c_uint(form), st)
                    assert rc == 0
                res[name] = {'behind_standin_us': round(timed(it) - base, 2), 'back_to_back_us': round(timed(b2b), 2)}
        out[f'n{sets[0].inst.N}'] = {'standin_us': round(base, 1), **res}
        print(f'n{sets[0].inst.N}', json.dumps(out[f'n{sets[0].inst.N}']), flush=True)


if __name__ == '__main__':
    main()
