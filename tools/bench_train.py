"""Training-step timing (BASELINE.json configs[4] shape: 128 image pairs per GPU): phases of NativeTrainer.step
timed with HIP events on the launch stream.  Secondary measurement -- bench.py stays the headline metric.
    python tools/bench_train.py [--batch 128] [--steps 10] [--warmup 3]
"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd
from ffrnet_amd import synth, train


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--train-opt', action='append', default=[], metavar='NAME=INT', help='ffr_train_option before the timed steps; repeatable')
    ap.add_argument('--cpu-baseline', action='store_true', help='also time the oracle (torch CPU autograd) on a small batch')
    a = ap.parse_args()
    specs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g0_state_dict_keys.json')))
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(synth.synth_state_dict(specs['encoder'], seed=0))
    tr = ffrnet_amd.NativeTrainer(eng, synth.synth_state_dict(specs['recnet'], seed=0), lr=1e-3)
    for kv in a.train_opt:
        eng.train_option(kv.split('=')[0], int(kv.split('=')[1]))
    non, ocl, label = synth.synth_train_batch(a.batch, seed=11)
    non, ocl, label = non.cuda(), ocl.cuda(), label.cuda()
    n = a.batch
    ev = lambda: torch.cuda.Event(enable_timing=True)
    phases = ['encoder', 'recnet_fwd', 'losses', 'recnet_bwd', 'adam']
    tot = {p: 0.0 for p in phases}
    wall = 0.0
    for it in range(a.warmup + a.steps):
        e = [ev() for _ in range(6)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e[0].record()
        with torch.no_grad():
            fm, f_enc = eng.encoder_forward(torch.cat((non, ocl), 0))
        e[1].record()
        lab2 = torch.cat((label, label))
        eng.train_forward(fm, lab2, groups=2, want=())
        e[2].record()
        out5 = eng.train_losses(f_enc)
        e[3].record()
        eng.train_zero_grad()
        eng.train_backward_losses()
        e[4].record()
        eng.train_adam_step(1e-3, (0.9, 0.999), 1e-8, 0.0, 1.0)
        e[5].record()
        torch.cuda.synchronize()
        if it >= a.warmup:
            wall += time.perf_counter() - t0
            for i, p in enumerate(phases):
                tot[p] += e[i].elapsed_time(e[i + 1])
    ms = {p: round(v / a.steps, 3) for p, v in tot.items()}
    # the product path: NativeTrainer.step = ffr_train_iteration (one launch-only call) + clip/Adam
    for it in range(3):
        tr.step(non, ocl, label)          # warm-up of the one-call path 
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.steps):
        items = tr.step(non, ocl, label)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / a.steps * 1e3
    # algorithmic work (SURVEY 8d counting: direct convolutions, 2*MACs): encoder 12.5934 GFLOP/image forward only,
    # RecNet 2.5493 GFLOP/image forward and twice that backward (data + weight gradients)
    gflop = 2 * n * (12.5934 + 3 * 2.5493)
    # not a roofline fraction: the Winograd layers execute 1/4 .. 1/3 of these multiplies (bench.py reports executed FLOPs)
    roof = {'effective_tflops_algorithmic': round(gflop / step_ms, 2), 'gflop_per_iteration_algorithmic': round(gflop, 1),
            'note': 'algorithmic direct-convolution FLOPs per iteration / wall time; NOT comparable with the 157.3 TFLOP/s '
                    'MFMA peak because Winograd executes fewer multiplies'}
    cpu = None
    if a.cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import ffr_oracle_train as OT
        nb = 8
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        sd_e = synth.synth_state_dict(specs['encoder'], seed=0)
        sd_r = synth.synth_state_dict(specs['recnet'], seed=0)
        cn, co, cl = synth.synth_train_batch(nb, seed=11)
        opt = OT.new_adam_state(sd_r)
        OT.train_step(sd_e, sd_r, opt, cn, co, cl, lr=1e-3)
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            OT.train_step(sd_e, sd_r, opt, cn, co, cl, lr=1e-3)
        dt = (time.perf_counter() - t0) / reps
        cpu = {'value': round(nb / dt, 2), 'unit': 'pairs/s', 'cores': torch.get_num_threads(), 'kind': 'port',
               'sample': '%d pairs per iteration, %d iterations, torch %s CPU autograd (oracle/ffr_oracle_train.py)' % (nb, reps, torch.__version__)}
    print(json.dumps({'work': roof, 'cpu_baseline': cpu, 'metric': 'RecNet training iterations/s (encoder frozen; clean+occluded pairs)', 'batch_pairs_per_gpu': n,
                      'ms_per_step': round(step_ms, 3), 'pairs_per_s': round(n / step_ms * 1e3, 1), 'phase_ms': ms,
                      'phased_ms_per_step': round(wall / a.steps * 1e3, 3), 'losses': [round(float(l), 5) for l in items]}))


if __name__ == '__main__':
    main()
