#!/usr/bin/env python3
"""Benchmark of the FFR-Net embedding path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of synthetic input per GPU:
    x[256,3,112,112] fp32 (resident in HBM) -> encoder + RecNet (HIP kernels) -> f_new, f
    -> (N > 1) RCCL all-gather of the 512-d embeddings over xGMI
    -> pairwise cosine scores (lfw/lfw_eval.py:246,248) on the gathered embeddings.
Workload = BASELINE.json configs[2] (IR-SE50 + RecBlock forward, batch 256, fp32), which
contains configs[1] (backbone only).  Resolution: the reference's live path is 112x112;
a 112x96 input cannot produce an embedding in the reference.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the fp32-MFMA implicit-GEMM
convolution, timed with hipEvents on the launch stream) and `cpu_baseline` (the oracle =
stock-torch CPU restatement of the reference, timed on this host's cores; rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import ffrnet_amd  # noqa: E402
from ffrnet_amd import synth  # noqa: E402

GFLOP_PER_IMAGE = 15.1427          # 2*MACs of every conv/linear/bmm, 112x112
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def state_dict_specs():
    enc = ffrnet_amd.Backbone(num_layers=50, drop_ratio=0.6, mode='ir_se')
    rec = ffrnet_amd.RecNet()
    return ({k: tuple(v.shape) for k, v in enc.state_dict().items()},
            {k: tuple(v.shape) for k, v in rec.state_dict().items()})


def pmc_traffic():
    """HBM bytes per k_igemm launch from the committed rocprofv3 --pmc passes (tools/pmc_bench.sh:
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH doubled as MI355X_MICROARCH.md prescribes for
    16-B-per-lane streams on gfx950).  PMC counters cannot be read from inside this process."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_traffic.json')) as f:
            ig = json.load(f)['igemm']
        return {'hbm_bytes_per_launch': int(ig['hbm_bytes_per_launch_corrected']),
                'source': 'profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, %d launches)'
                          % ig['launches']}
    except Exception:
        return None


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs but grants 16: 256 torch threads there ran the
    oracle 100x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(sd_e, sd_r, budget_s=12.0):
    """Oracle (kind 'port') on the host cores: batches of 8 images (BASELINE configs[0])
    until ~budget_s of CPU work, after one warm-up batch."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ffr_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    x = synth.synth_images(8, seed=123)
    O.embed(sd_e, sd_r, x)
    n, t0 = 0, time.perf_counter()
    while True:
        O.embed(sd_e, sd_r, x)
        n += 8
        dt = time.perf_counter() - t0
        if dt > budget_s:
            break
    return {'value': round(n / dt, 2), 'unit': 'embeddings/s', 'cores': torch.get_num_threads(),
            'kind': 'port',
            'sample': '%d images in batches of 8 (configs[0] shape), %.1f s, torch %s CPU, %d threads '
                      '(%d logical CPUs visible)' % (n, dt, torch.__version__, torch.get_num_threads(),
                                                     os.cpu_count() or 0)}


def train_workload(args, world, rank, local, dist):
    """Secondary workload (SURVEY 8 row N3 / BASELINE configs[4]): whole training iterations through NativeTrainer."""
    dev = torch.device('cuda', local)
    spec_e, spec_r = state_dict_specs()
    eng = ffrnet_amd.Engine(local)
    eng.load_encoder(synth.synth_state_dict(spec_e))
    tr = ffrnet_amd.NativeTrainer(eng, synth.synth_state_dict(spec_r), lr=1e-3)
    tr.broadcast_params(0)
    B = args.batch if args.batch != 256 else 128           # pairs per GPU (1024 pairs over 8 GPUs)
    non, ocl, label = synth.synth_train_batch(B, seed=500 + rank)
    non, ocl, label = non.to(dev), ocl.to(dev), label.to(dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step(non, ocl, label)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        items = tr.step(non, ocl, label)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert all(torch.isfinite(l) for l in items)
    if rank == 0:
        print(json.dumps({
            'metric': 'training image pairs/sec (frozen IR-SE50 encoder + RecNet forward/backward + CosFace head + clip + Adam)',
            'value': round(world * B * args.steps / dt, 1), 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[4]: training step, %d pairs per GPU, 112x112x3 fp32' % B,
                       'pairs_per_gpu': B, 'global_pairs': world * B,
                       'parallelism': 'data parallel x%d, one RCCL all-reduce of the flat fp32 gradient buffer per step' % world},
            'roofline': None, 'cpu_baseline': None}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='images per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--workload', choices=('embed', 'train'), default='embed',
                    help="embed (default): BASELINE.json's metric; train: the RecNet training iteration of configs[4] "
                         '(128 image pairs per GPU unless --batch is given), data parallel with one all-reduce of the '
                         'flat gradient buffer')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run '
                             '--nproc-per-node %d' % (args.gpus, args.gpus))
    dist = None
    # FFR_BENCH_BACKEND=gloo FFR_BENCH_ONE_DEVICE=1: every rank on cuda:0 over gloo -- exercises the N > 1 code path
    # on a 1-GPU box (RCCL refuses two ranks on one device); timings of such a run mean nothing
    backend = os.environ.get('FFR_BENCH_BACKEND', 'nccl')
    if os.environ.get('FFR_BENCH_ONE_DEVICE') == '1':
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device('cuda', local)
    if args.workload == 'train':
        return train_workload(args, world, rank, local, dist)

    spec_e, spec_r = state_dict_specs()
    sd_e = synth.synth_state_dict(spec_e)
    sd_r = synth.synth_state_dict(spec_r)
    eng = ffrnet_amd.Engine(local)
    eng.load_encoder(sd_e)
    eng.load_recnet(sd_r)
    B = args.batch
    eng.reserve(B)
    x = synth.synth_images(B, seed=124 + rank).to(dev)
    f_new = torch.empty((B, 512), device=dev)
    f = torch.empty((B, 512), device=dev)
    g_new = torch.empty((world * B, 512), device=dev) if world > 1 else f_new
    g_old = torch.empty((world * B, 512), device=dev) if world > 1 else f

    def step():
        eng.embed(x, out=(f_new, f))
        if world > 1:
            dist.all_gather_into_tensor(g_new, f_new)
            dist.all_gather_into_tensor(g_old, f)
        # image 2i / 2i+1 of the gathered batch form verification pair i
        a, b = g_new.view(-1, 2, 512)[:, 0], g_new.view(-1, 2, 512)[:, 1]
        return eng.cosine_scores(a.contiguous(), b.contiguous())

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scores = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert torch.isfinite(scores).all()
    value = world * B * args.steps / dt

    roof = None
    if rank == 0 and not args.no_roofline:
        eng.profile_enable(True)
        nprof = 3
        for _ in range(nprof):
            eng.embed(x, out=(f_new, f))
        torch.cuda.synchronize()
        st = eng.profile_read()
        eng.profile_enable(False)
        c, wn = st['conv_igemm'], st['wino']
        tot_ms = sum(v['ms'] for v in st.values())
        conv_ms = c['ms'] + wn['ms']                     # GEMM launches + Winograd transforms
        ach = c['flops'] / (conv_ms * 1e-3) / 1e12       # ALGORITHMIC (direct-conv) FLOPs, SURVEY 8(d)
        mfma = c['flops_executed'] / (c['ms'] * 1e-3) / 1e12
        roof = {'bound': 'mfma',
                'kernel': 'k_igemm / k_gemm_stream (fp32 MFMA implicit GEMM: direct convs, FC, and the 36 batched GEMMs of '
                          'every Winograd F(4x4,3x3) conv) + the k_wino_in/k_wino_out transform launches',
                'achieved': round(ach, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': pmc_traffic(),
                'note': 'achieved = algorithmic direct-convolution FLOPs / (k_igemm + transform time); > 1.0 of the '
                        'MFMA peak is possible because Winograd executes 1/4 of the multiplies',
                'mfma_executed_tflops': round(mfma, 2),
                'mfma_executed_frac': round(mfma / PEAK_FP32_MFMA_TFLOPS, 4),
                'launches_per_step': c['launches'] // nprof,
                'avg_launch_us': round(c['ms'] * 1e3 / max(1, c['launches']), 2),
                'gflop_per_launch': round(c['flops'] / max(1, c['launches']) / 1e9, 3),
                'gflop_executed_per_launch': round(c['flops_executed'] / max(1, c['launches']) / 1e9, 3),
                'kernel_ms_per_step': {k: round(v['ms'] / nprof, 3) for k, v in st.items() if v['launches']},
                'all_kernels_ms_per_step': round(tot_ms / nprof, 3),
                'whole_path_achieved': round(value / world * GFLOP_PER_IMAGE / 1e3, 2),
                'whole_path_frac': round(value / world * GFLOP_PER_IMAGE / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4)}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd_e, sd_r)

    if rank == 0:
        out = {'metric': 'face embeddings/sec (IR-SE50 + RecNet forward, 112x112 fp32) at batch 256 per GPU',
               'value': round(value, 1), 'unit': 'embeddings/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
               'data': 'synthetic',
               'config': {'workload': 'configs[2]: IR-SE50 + RecBlock (spatial+channel) forward, batch %d per GPU, '
                                      '112x112x3 fp32 -> f_new,f [512]; + all-gather of embeddings and pair '
                                      'cosine scores (112x96 of the BASELINE label cannot be embedded by the '
                                      'reference: BASELINE.md)' % B,
                          'batch_per_gpu': B, 'global_batch': world * B, 'gflop_per_image': GFLOP_PER_IMAGE,
                          'parallelism': 'image-sharded x%d, RCCL all-gather of embeddings' % world},
               'roofline': roof, 'cpu_baseline': cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
