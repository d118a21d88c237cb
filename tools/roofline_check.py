#!/usr/bin/env python3
"""Recompute bench.py's roofline block for the dominant kernel from the committed rocprofv3 summary and a layer table.
    python tools/roofline_check.py profiles/r05_bench_batch256_kernel_stats.csv profiles/r05_bench.json [--tol 0.03]
    python tools/roofline_check.py --train profiles/r05_train_step_kernel_stats.csv profiles/r05_train_step.json
Independent of the library: executed / useful FLOPs come from the layer shapes below (batch from the bench line),
the average launch time from the rocprofv3 --kernel-trace --stats CSV.  Exits non-zero when a figure of the bench
line disagrees by more than --tol.  (The bench line and the CSV are two runs of the same command on the same build.)"""
import csv
import json
import math
import sys

PEAK = 157.3e12
GFLOP_PER_IMAGE = 15.1427


def fused_layers():
    """(H, cin, cout) of every convolution k_wino_fused runs at batch >= 128: the 3x3 / stride-1 convolutions of the
    trunk (model_ir_se50.py:66-69: conv1 of 24 units at the unit's input size, conv2 of the 20 stride-1 units) and of
    RecNet: Conv4Space's first three, ChannelFlipMerge and Conv4Merge (recnet.py:362-394)."""
    out, h, cin = [], 112, 64
    for depth, n in ((64, 3), (128, 4), (256, 14), (512, 3)):
        for u in range(n):
            out.append((h, cin, depth))                  # conv1: stride 1 at the input resolution
            if u == 0:
                h //= 2                                  # conv2 of a stage's first unit has stride 2: direct kernel
            else:
                out.append((h, depth, depth))
            cin = depth
    out += [(7, 561, 256), (7, 256, 256), (7, 256, 256)]
    out += [(7, 1024, 512), (7, 512, 512), (7, 512, 512), (7, 1536, 512), (7, 512, 512), (7, 512, 512)]
    return out


def pad(v, m):
    return (v + m - 1) // m * m


def check_train(stats, bench, tol):
    """--train: the training workload.  Independent figures from the rocprofv3 summary of `bench.py --workload train
    --steps 5 --warmup 2 --no-roofline` (7 iterations = 7 k_stem launches): launches of the fused Winograd kernels per iteration, of
    k_wgrad per iteration, and the pure kernel time of both classes, against `roofline.per_class` of the bench line (whose times are
    hipEvent pairs that also contain the kernel boundary in front of each launch: they may exceed the CSV's by up to 8 %)."""
    line = [l for l in open(bench) if l.startswith('{')][-1]
    pc = json.loads(line)['roofline']['per_class']
    rows = list(csv.DictReader(open(stats)))
    iters = sum(int(r['Calls']) for r in rows if 'k_stem' in r['Name'])
    bad = 0
    print('%d iterations in %s' % (iters, stats))
    print('%-22s %14s %14s %14s %14s' % ('class', 'launches (csv)', 'launches (line)', 'ms (csv)', 'ms (line)'))
    # (class, kernels counted as its launches, kernels whose time belongs to it: a k_wgrad scope also holds its split-K reduction)
    for cls, pat, tpat, slack in (('wino_fused', 'k_wino_fused', 'k_wino_fused', 0.08), ('wgrad', 'k_wgrad<', 'k_wgrad', 0.08)):
        sel = [r for r in rows if pat in r['Name']]
        calls = sum(int(r['Calls']) for r in sel)
        ms = sum(int(r['TotalDurationNs']) for r in rows if tpat in r['Name']) / iters / 1e6
        lp, mp = pc[cls]['launches_per_step'], pc[cls]['ms_per_step']
        ok = calls == lp * iters and 0 <= (mp - ms) / ms <= max(tol, slack)
        bad += not ok
        print('%-22s %14.1f %14d %14.3f %14.3f%s' % (cls, calls / iters, lp, ms, mp, '' if ok else '   <-- DISAGREES'))
    sys.exit(1 if bad else 0)


def main():
    tol = float(sys.argv[sys.argv.index('--tol') + 1]) if '--tol' in sys.argv else 0.03
    if '--train' in sys.argv:
        args = [a for a in sys.argv[1:] if not a.startswith('--') and a != str(tol)]
        return check_train(args[0], args[1], tol)
    stats, bench = sys.argv[1], sys.argv[2]
    line = [l for l in open(bench) if l.startswith('{')][-1]
    b = json.loads(line)
    r = b['roofline']
    n = b['config']['batch_per_gpu']
    calls = ns = 0
    for row in csv.DictReader(open(stats)):
        if 'k_wino_fused' in row['Name']:          # k_wino_fused<.,.>, k_wino_fused_mixed
            calls += int(row['Calls'])
            ns += int(row['TotalDurationNs'])
    avg_us = ns / calls / 1e3
    L = fused_layers()
    ex = us = 0.0
    f44 = sum(2 * n * h * h * 9 * cin * cout / 4 for h, cin, cout in L)      # direct-convolution FLOPs / 4: all-F(4x4), no padding
    for h, cin, cout in L:
        if h == 14 and cin == 256 and n * 16 // 32 * (pad(cout, 64) // 64) >= 512:
            # exact 4+4+3+3 tiling (wino_mixed.hip): four tile types, 4 tiles each per image, 36 / 30 / 30 / 25 xi padded to 36 / 32 / 32 / 28
            for x, xp in ((36, 36), (30, 32), (30, 32), (25, 28)):
                ex += 2 * xp * pad(4 * n, 32) * pad(cin, 32) * pad(cout, 64)
                us += 2 * x * 4 * n * cin * cout
            continue
        t = n * math.ceil(h / 4) ** 2
        ex += 2 * 36 * pad(t, 32) * pad(cin, 32) * pad(cout, 64)
        us += 2 * n * h * h * 9 * cin * cout / 4
    nl = len(L)
    mine = {'launches_per_step': nl, 'avg_launch_us': avg_us,
            'gflop_executed_per_launch': ex / nl / 1e9, 'gflop_useful_per_launch': us / nl / 1e9,
            'frac': ex / nl / (avg_us * 1e-6) / PEAK, 'frac_useful': us / nl / (avg_us * 1e-6) / PEAK,
            'frac_useful_f4x4_equivalent': f44 / nl / (avg_us * 1e-6) / PEAK,
            'frac_algorithmic_survey_8d': b['value'] / b['n_gpus'] * GFLOP_PER_IMAGE * 1e9 / PEAK}
    bad = 0
    print('%-30s %12s %12s %8s' % ('figure', 'recomputed', 'bench line', 'diff'))
    for k, v in mine.items():
        got = r.get(k)
        d = abs(got - v) / abs(v) if got is not None else float('inf')
        flag = '' if d <= tol else '   <-- DISAGREES'
        bad += d > tol
        print('%-30s %12.4f %12s %7.2f%%%s' % (k, v, got, d * 100, flag))
    print('k_wino_fused: %d calls in %s = %d forwards of %d launches' % (calls, stats, calls // nl, nl))
    if calls % nl:
        print('   <-- call count is not a multiple of %d launches per step' % nl)
        bad += 1
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
