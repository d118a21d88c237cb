#!/usr/bin/env python3
"""Benchmark of the FFR-Net embedding path on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N > 1 from a plain shell: this process spawns one child per GPU (it never touches the GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        one rank per GPU over RCCL (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment)

One step = one pass of the hot path over one batch of synthetic input per GPU:
    x[B,3,112,112] fp32 (resident in HBM) -> encoder + RecNet (HIP kernels) -> f_new, f
    -> (N > 1) RCCL all-gather of the 512-d embeddings over xGMI
    -> pairwise cosine scores (lfw/lfw_eval.py:246,248) on the gathered embeddings.
Workload = BASELINE.json configs[2] (IR-SE50 + RecBlock forward, batch 256 per GPU, fp32, weak scaling);
`--pairs-per-step P` runs configs[3]'s shape instead: P verification pairs per step split over the GPUs
(2P/N images per GPU, strong scaling).  Resolution: the reference's live path is 112x112; a 112x96 input
cannot produce an embedding in the reference -- the 112x96 TRUNK is timed as a
secondary line.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel (k_wino_fused: the fp32-MFMA
GEMMs + output transform of every Winograd convolution) with EXECUTED FLOPs over hipEvent time on the
launch stream, so `frac` <= 1; `cpu_baseline` is the oracle (stock-torch CPU restatement of the
reference) timed on this host's cores (rank 0, N = 1).  Before anything is timed the 8 images of golden
G1 are embedded and compared with the reference's own outputs (`parity_checked`); the run fails if the
error exceeds the 1e-3 contract.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE = 15.1427          # 2*MACs of every conv/linear/bmm, 112x112
GFLOP_TRUNK_112 = 12.5677          # trunk only (input_layer -> body -> bn), 112x112
GFLOP_TRUNK_96 = 10.7724           # trunk only, 112x96
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz
PEAK_HBM_TBS = 8.0                 # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s measured achievable)
PARITY_TOL = 1e-3                  # BASELINE.json north_star


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=256, help='images per GPU per step (weak scaling)')
    ap.add_argument('--pairs-per-step', type=int, default=0,
                    help='> 0: strong scaling, this many verification pairs per step over all GPUs '
                         '(BASELINE configs[3]: 512 pairs -> 128 images per GPU at 8 GPUs)')
    ap.add_argument('--opt', action='append', default=[], metavar='NAME=INT',
                    help='experiment knob of the native handle (ffr_set_option, DESIGN.md 3.3); repeatable')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true')
    ap.add_argument('--strong-pairs', type=int, default=512,
                    help='N > 1 only: pairs per step of the configs[3] strong-scaling measurement appended as a `secondary` entry of the '
                         'default weak-scaling line (split over the ranks; 0: off)')
    ap.add_argument('--workload', choices=('embed', 'train'), default='embed',
                    help="embed (default): BASELINE.json's metric; train: the RecNet training iteration of configs[4] "
                         '(128 image pairs per GPU unless --batch is given), data parallel with one all-reduce of the '
                         'flat gradient buffer')
    return ap.parse_args()


# ---- self launch ------------------------------------------------------------------------------------
def self_launch(args):
    """`python bench.py --gpus N` from a plain shell: one child process per GPU.  The parent imports nothing
    that initialises the GPU; rank 0's JSON line is passed through."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies must not leave the others waiting at the rendezvous (or in a collective) until a timeout: the
    # first non-zero exit ends the job
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                rc = rc or r
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=10)
    sys.stdout.write(b''.join(out).decode())
    sys.stdout.flush()
    return rc


# ---- helpers ----------------------------------------------------------------------------------------
def state_dict_specs():
    import ffrnet_amd
    enc = ffrnet_amd.Backbone(num_layers=50, drop_ratio=0.6, mode='ir_se')
    rec = ffrnet_amd.RecNet()
    return ({k: tuple(v.shape) for k, v in enc.state_dict().items()},
            {k: tuple(v.shape) for k, v in rec.state_dict().items()})


def so_sha256():
    from ffrnet_amd import native
    h = hashlib.sha256()
    with open(native.lib_path(), 'rb') as f:
        h.update(f.read())
    return h.hexdigest()


def under_profiler():
    """True when this process runs under rocprofv3 / rocprof (its tool library is preloaded or its environment set)."""
    if any(k.startswith(('ROCP_', 'ROCPROF', 'ROCPROFILER_')) for k in os.environ):
        return True
    return any('rocprof' in os.environ.get(k, '') for k in ('LD_PRELOAD', 'HSA_TOOLS_LIB'))


def _pmc_fields(d, src):
    return {'hbm_bytes_per_launch': d['dominant']['hbm_bytes_per_launch'],
            'launches_per_step': d['dominant']['launches_per_step'],
            'hbm_gb_per_step_all_kernels': d['gb_per_step'],
            'mfma_busy_frac_of_kernel_time': d['dominant'].get('mfma_busy_frac_of_kernel_time'),
            'batch': d.get('batch', 256), 'source': src}


def pmc_traffic(sha, batch=256, live=True, workload='embed'):
    """HBM bytes per launch of the dominant kernel and per step from rocprofv3 --pmc passes (tools/pmc_bench.sh:
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH doubled as MI355X_MICROARCH.md prescribes for 16-B-per-lane
    streams on gfx950).  PMC counters cannot be read from inside this process: the committed summary of the round
    (profiles/r*_pmc_hbm_traffic.json) carries the sha256 of the library AND the batch it was measured with and is used
    when both match this run; otherwise the passes are run now, as child processes in their own process group (three
    passes, bounded below the timeout used here; the whole group is killed on a timeout so that no profiler pass keeps
    the GPU busy under the measurements that follow), with the profiler's own environment stripped.  Never nested:
    when this process itself runs under rocprofv3 the figure is reported as absent instead."""
    import glob
    pat = 'r*_pmc_hbm_traffic.json' if workload == 'embed' else 'r*_pmc_%s_traffic.json' % workload
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', pat)), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        if d.get('so_sha256') == sha and 'dominant' in d and d.get('batch', 256) == batch and d.get('workload', 'embed') == workload:
            rel = os.path.relpath(path, ROOT)
            return _pmc_fields(d, '%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same build: sha256 %s..., batch %d)'
                               % (rel, sha[:12], batch))
    import shutil
    import signal
    why = 'live passes disabled'
    if live and shutil.which('rocprofv3') and os.environ.get('FFR_BENCH_LIVE_PMC', '1') != '0' and not under_profiler():
        t0 = time.perf_counter()
        env = {k: v for k, v in os.environ.items()
               if not k.startswith(('ROCP_', 'ROCPROF', 'ROCPROFILER_')) and k not in ('LD_PRELOAD', 'HSA_TOOLS_LIB',
                                                                                        'RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
        env.update(GRAFT_REPO_ROOT=ROOT, FFR_PMC_BATCH=str(batch), FFR_PMC_PASS_TIMEOUT='110', FFR_BENCH_LIVE_PMC='0',
                   FFR_PMC_WORKLOAD=workload)
        proc = None
        try:
            proc = subprocess.Popen(['bash', os.path.join(ROOT, 'tools', 'pmc_bench.sh')], env=env, cwd=ROOT,
                                    stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            rc = proc.wait(timeout=380)                 # 3 passes x 110 s + the summary < 380 s
            if rc != 0:
                raise RuntimeError('tools/pmc_bench.sh exited with %d' % rc)
            with open(os.path.join(ROOT, 'gpurun_out', 'pmc_bench', 'summary.json')) as f:
                d = json.load(f)
            if d.get('so_sha256') == sha and 'dominant' in d and d.get('batch', 256) == batch and d.get('workload', 'embed') == workload:
                return _pmc_fields(d, 'measured in this run: tools/pmc_bench.sh as child processes (rocprofv3 --pmc, %.0f s); '
                                      'no committed summary matches this build (sha256 %s...) and batch %d'
                                   % (time.perf_counter() - t0, sha[:12], batch))
            why = 'the live passes measured another build or batch'
        except Exception as e:          # no profiler, no permission, timeout: say so instead of inventing a number
            if proc is not None and proc.poll() is None:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)         # bash, timeout, rocprofv3 and its python child
                except Exception:
                    pass
                try:
                    proc.wait(timeout=30)
                except Exception:
                    pass
            why = 'the live rocprofv3 --pmc passes failed: %s' % str(e)[:200]
    elif under_profiler():
        why = 'this run is itself under a profiler (no nested rocprofv3)'
    return {'hbm_bytes_per_launch': None,
            'note': 'no profiles/r*_pmc_hbm_traffic.json was measured with this build of libffrnet_hip.so (sha256 %s...) at '
                    'batch %d; %s: re-run tools/pmc_bench.sh' % (sha[:12], batch, why)}


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


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(sd_e, sd_r, budget_s=12.0):
    """Oracle (kind 'port') on the host cores.  Headline: batches of 8 images (BASELINE configs[0]) for ~budget_s
    after one warm-up batch, all granted cores.  Also: one thread at batch 8 and all cores at batch
    256, 1 warm-up + min of 3."""
    import torch
    from ffrnet_amd import synth
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
    head = n / dt

    def min_of_3(xb):
        O.embed(sd_e, sd_r, xb)
        best = 1e30
        for _ in range(3):
            t = time.perf_counter()
            O.embed(sd_e, sd_r, xb)
            best = min(best, time.perf_counter() - t)
        return xb.size(0) / best

    b256 = min_of_3(synth.synth_images(256, seed=124))
    torch.set_num_threads(1)
    one = min_of_3(x)
    torch.set_num_threads(cores)
    return {'value': round(head, 2), 'unit': 'embeddings/s', 'cores': cores, 'kind': 'port',
            'sample': '%d images in batches of 8 (configs[0] shape), %.1f s, torch %s CPU, %d threads (%d logical CPUs visible)'
                      % (n, dt, torch.__version__, cores, os.cpu_count() or 0),
            'cpu_model': cpu_model(),
            'batch256_all_cores': round(b256, 2), 'batch8_one_thread': round(one, 2),
            'note': 'batch256_all_cores / batch8_one_thread: 1 warm-up + min of 3 passes'}


def percentiles(ms):
    s = sorted(ms)

    def q(p):
        return s[min(len(s) - 1, int(round(p * (len(s) - 1))))]
    return {'median': round(q(0.5), 3), 'p10': round(q(0.1), 3), 'p90': round(q(0.9), 3), 'n': len(s)}


def timed_events(fn, warm, reps):
    """hipEvent pairs on the current stream around every call of fn -> list of ms."""
    import torch
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


def lfw_protocol_line(eng, dev):
    """configs[3] end to end on one GPU: 6000 pairs (12000 images resident in HBM, pair batches of 512) through the
    product harness -- embed, ffr_cosine_scores, ffr_lfw_fold_accuracy for f_new and f -- wall clock incl. the
    device->host copies of the results."""
    import torch
    from ffrnet_amd import lfw
    n, bs = 6000, 512
    g = torch.Generator(device=dev).manual_seed(9)
    loader = []
    for s0 in range(0, n, bs):
        m = min(bs, n - s0)
        a = torch.rand((m, 3, 112, 112), device=dev, generator=g) * 2 - 1
        b = torch.rand((m, 3, 112, 112), device=dev, generator=g) * 2 - 1
        lab = ((torch.arange(s0, s0 + m) % 600) < 300).long()
        mix = lab.view(-1, 1, 1, 1).to(dev).float() * 0.7
        loader.append(dict(img1=a, img2=mix * a + (1 - mix) * b, label=lab, idx=torch.arange(s0, s0 + m)))
    lfw.get_avg_accuracy(eng.embed, loader[:1], n_folds=2)            # warm-up (arena for 1024 images)
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(2):
        t0 = time.perf_counter()
        acc_new, acc = lfw.get_avg_accuracy(eng.embed, loader)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return {'workload': 'configs[3] on 1 GPU: LFW protocol, 6000 synthetic pairs in pair batches of 512 '
                        '(12000 images resident in HBM) -> embeddings -> cosine scores -> 10-fold threshold protocol '
                        'for f_new and f, all on the device (ffrnet_amd.lfw.get_avg_accuracy, default path)',
            'value': round(n / best, 1), 'unit': 'pairs/s', 'seconds': round(best, 4),
            'embeddings_per_s': round(2 * n / best, 1), 'acc_new': acc_new, 'acc': acc,
            'note': 'random pairs generated on the device: the accuracies are not comparable with golden G9 '
                    '(tests/test_gpu_parity.py::test_lfw_protocol_6000_pairs_matches_reference holds that)'}


def strong_scaling_line(eng, dist, world, rank, dev, steps, warmup, pairs):
    """BASELINE configs[3] on N GPUs: `pairs` verification pairs per step (lfw/lfw_eval.py:226-252 batches) SPLIT over
    the ranks -- 2 * pairs / world images per GPU, one packed all-gather of the embeddings, pair scores -- timed like
    the main line (barrier + synchronize on both sides, MAX over ranks), inputs resident in HBM.  Then ONE pass of two
    such pair batches from HOST memory through the product harness (ffrnet_amd.lfw.calculate_distance -> ShardFeeder),
    which says what each rank copies over PCIe.  Every rank calls this; rank 0 gets the entry."""
    import torch
    from ffrnet_amd import synth, lfw
    if pairs <= 0 or (2 * pairs) % world:
        return None
    B = 2 * pairs // world
    eng.reserve(max(B, 8))
    x = synth.synth_images(B, seed=324 + rank).to(dev)
    pack = torch.empty((2, B, 512), device=dev)
    gathered = torch.empty((world, 2, B, 512), device=dev)
    coll_ev = []

    def step(timed=False):
        eng.embed(x, out=(pack[0], pack[1]))
        if timed:
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
        dist.all_gather_into_tensor(gathered.view(world * 2 * B, 512), pack.view(2 * B, 512))
        if timed:
            c1.record()
            coll_ev.append((c0, c1))
        pr = gathered[:, 0].reshape(-1, 2, 512)
        return eng.cosine_scores(pr[:, 0].contiguous(), pr[:, 1].contiguous())

    def fence():
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        scores = step(timed=True)
    fence()
    dt_local = time.perf_counter() - t0
    assert torch.isfinite(scores).all() and scores.numel() == pairs
    assert torch.equal(gathered[rank], pack), 'all-gather: rank %d does not find its own embeddings' % rank
    # the same shape from host memory through the harness: two pair batches, every rank feeds ITS shard only
    i1, i2, lab = synth.synth_pairs(2 * pairs, seed=31, block=max(2, pairs // 5 * 2))
    loader = [dict(img1=i1[s0:s0 + pairs], img2=i2[s0:s0 + pairs], label=lab[s0:s0 + pairs], idx=torch.arange(s0, s0 + pairs))
              for s0 in (0, pairs)]
    fence()
    t0 = time.perf_counter()
    pn, p = lfw.calculate_distance(loader, eng.embed)
    fence()
    dt_host = time.perf_counter() - t0
    st = lfw.last_feed_stats
    assert pn.shape == (2 * pairs, 3) and st['batches'] == 2
    mine = torch.tensor([dt_local / steps * 1e3, percentiles([a.elapsed_time(b) for a, b in coll_ev])['median'], dt_host,
                         float(st['h2d_bytes']), float(st['shard_bytes']), float(st['full_batch_bytes'])], device=dev, dtype=torch.float64)
    flat = torch.empty(world * 6, device=dev, dtype=torch.float64)
    dist.all_gather_into_tensor(flat, mine)
    allr = flat.view(world, 6)
    ms = allr[:, 0].max().item()
    host_s = allr[:, 2].max().item()
    return {'workload': 'configs[3] on %d GPUs, STRONG scaling: %d verification pairs per step split over the ranks = %d images per '
                        'GPU, one packed all-gather of the embeddings, pair cosine scores (inputs resident in HBM)' % (world, pairs, B),
            'scaling': 'strong', 'value': round(2 * pairs / ms * 1e3, 1), 'unit': 'embeddings/s', 'pairs_per_s': round(pairs / ms * 1e3, 1),
            'ms_per_step': round(ms, 3), 'steps': steps, 'warmup': warmup, 'n_gpus': world, 'batch_per_gpu': B,
            'per_rank': {'wall_ms_per_step': [round(v, 3) for v in allr[:, 0].tolist()],
                         'all_gather_ms_hipevents_median': [round(v, 4) for v in allr[:, 1].tolist()],
                         'all_gather_bytes_per_rank': 2 * B * 512 * 4},
            'from_host_memory': {'what': 'two pair batches of %d pairs from host tensors through ffrnet_amd.lfw.calculate_distance '
                                         '(ShardFeeder: shard on the host, pinned staging, copy of the next batch under this batch\'s '
                                         'kernels), PCIe-inclusive' % pairs,
                                 'pairs_per_s': round(2 * pairs / host_s, 1), 'seconds': round(host_s, 4),
                                 'h2d_bytes_per_rank': [int(v) for v in allr[:, 3].tolist()],
                                 'shard_bytes_per_rank': [int(v) for v in allr[:, 4].tolist()],
                                 'full_batch_bytes': int(allr[0, 5].item())}}


def cpu_baseline_train(budget_s=15.0):
    """The training iteration of the oracle (oracle/ffr_oracle_train.py: stock-torch CPU autograd restatement of
    models/trainer.py:139-187) on the host cores: 8 pairs per iteration, one warm-up iteration, then whole iterations
    for about budget_s."""
    import torch
    from ffrnet_amd import synth
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ffr_oracle_train as OT
    cores = host_cores()
    torch.set_num_threads(cores)
    spec_e, spec_r = state_dict_specs()
    sd_e, sd_r = synth.synth_state_dict(spec_e), synth.synth_state_dict(spec_r)
    nb = 8
    cn, co, cl = synth.synth_train_batch(nb, seed=11)
    opt = OT.new_adam_state(sd_r)
    OT.train_step(sd_e, sd_r, opt, cn, co, cl, lr=1e-3)
    n, t0 = 0, time.perf_counter()
    while True:
        OT.train_step(sd_e, sd_r, opt, cn, co, cl, lr=1e-3)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s:
            break
    return {'value': round(n * nb / dt, 2), 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'sample': '%d iterations of %d pairs (clean + occluded image each), %.1f s, torch %s CPU autograd '
                      '(oracle/ffr_oracle_train.py), %d threads' % (n, nb, dt, torch.__version__, cores),
            'cpu_model': cpu_model()}


def train_roofline(eng, step_fn, B, fence, instrument=True, nprof=2, pmc=True):
    """`roofline` of the training iteration: executed FLOPs of its MFMA kernels over their hipEvent time on the launch
    stream -- k_wino_fused (encoder + RecNet forward + data gradients), k_igemm / k_gemm_stream (direct and batched-GEMM
    convolutions, linears) and k_wgrad (weight gradients as TN GEMMs) -- and every launch of the iteration in a kernel class."""
    if instrument:
        eng.profile_enable(True)
    for _ in range(nprof):
        step_fn()
    fence()
    if not instrument:
        return None
    st = eng.profile_read()
    eng.profile_enable(False)
    sha = so_sha256()
    cls = {k: {'ms_per_step': round(v['ms'] / nprof, 3), 'launches_per_step': v['launches'] // nprof,
               'executed_tflops': round(v['flops_executed'] / (v['ms'] * 1e-3) / 1e12, 2) if v['ms'] and v['flops_executed'] else None}
           for k, v in st.items() if v['launches']}
    dom = st['wino_fused']
    dom_s = dom['ms'] * 1e-3
    dom_tf = dom['flops_executed'] / dom_s / 1e12 if dom['ms'] else 0.0
    dom_useful_tf = dom['flops_useful'] / dom_s / 1e12 if dom['ms'] else 0.0
    mf = [st[k] for k in ('wino_fused', 'conv_igemm', 'wgrad')]
    mf_ms = sum(v['ms'] for v in mf)
    mf_tf = sum(v['flops_executed'] for v in mf) / (mf_ms * 1e-3) / 1e12 if mf_ms else 0.0
    mf_useful_tf = sum(v['flops_useful'] for v in mf) / (mf_ms * 1e-3) / 1e12 if mf_ms else 0.0
    tot_ms = sum(v['ms'] for v in st.values()) / nprof
    tr_pmc = pmc_traffic(sha, batch=B, live=False, workload='train') if pmc else {}
    return {'bound': 'mfma', 'kernel': 'k_wino_fused (frozen encoder + RecNet forward + data gradients)',
            'achieved': round(dom_tf, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(dom_tf / PEAK_FP32_MFMA_TFLOPS, 4),
            'frac_useful': round(dom_useful_tf / PEAK_FP32_MFMA_TFLOPS, 4),
            'traffic': tr_pmc.get('hbm_bytes_per_launch'), 'traffic_detail': tr_pmc,
            'avg_launch_us': round(dom['ms'] * 1e3 / max(1, dom['launches']), 2),
            'mfma_kernels': {'kernels': 'k_wino_fused + k_igemm/k_gemm_stream + k_wgrad', 'executed_tflops': round(mf_tf, 2),
                             'frac': round(mf_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                             'frac_useful': round(mf_useful_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                             'ms_per_step': round(mf_ms / nprof, 3)},
            'hbm_kernels_ms_per_step': round(tot_ms - mf_ms / nprof, 3),
            'all_kernels_ms_per_step': round(tot_ms, 3),
            'per_class': cls,
            'note': 'every launch of the iteration is instrumented (hipEvent pairs on the launch stream): the inference-path '
                    'classes (encoder), wgrad, and train_bn (BatchNorm statistics / apply / backward), train_loss (loss items, '
                    'CosFace head), train_optim (zero_grad, clip + Adam), train_xform (Winograd weight / gradient transforms, '
                    'dgrad packing, reflection folds), train_elem (remaining elementwise / layout kernels); the classes add up '
                    'to all_kernels_ms_per_step, which contains the dispatch gaps and is slightly more than ms_per_step'}


def train_step_line(eng, sd_r, dev, pairs=128, warm=2, iters=5):
    """SURVEY 8 row N3 / BASELINE configs[4] in the DEFAULT run: whole training iterations of
    models/trainer.py:139-187 through NativeTrainer on the engine the embedding benchmark just used (its frozen encoder is
    the one already loaded) -- frozen-encoder forward of clean + occluded images, RecNet train-mode forward, four losses,
    full backward, clip + Adam -- `warm` untimed + `iters` timed iterations of `pairs` image pairs, then two instrumented
    ones for the per-class times."""
    import torch
    import ffrnet_amd
    from ffrnet_amd import synth
    tr = ffrnet_amd.NativeTrainer(eng, sd_r, lr=1e-3)
    non, ocl, label = synth.synth_train_batch(pairs, seed=500)
    non, ocl, label = non.to(dev), ocl.to(dev), label.to(dev)
    for _ in range(warm):
        tr.step(non, ocl, label)
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        items = tr.step(non, ocl, label)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(torch.isfinite(l) for l in items)
    roof = train_roofline(eng, lambda: tr.step(non, ocl, label), pairs, torch.cuda.synchronize, pmc=False)
    return {'workload': 'configs[4] per-GPU shape on 1 GPU: RecNet training iteration (models/trainer.py:139-187), %d image pairs '
                        '= %d images: frozen IR-SE50 forward, RecNet train-mode forward, 4 losses + CosFace head, full backward, '
                        'clip + Adam (ffrnet_amd.NativeTrainer.step)' % (pairs, 2 * pairs),
            'value': round(pairs * iters / dt, 1), 'unit': 'pairs/s', 'ms_per_iteration': round(dt / iters * 1e3, 3),
            'ms_hipevents': percentiles([a.elapsed_time(b) for a, b in evs]), 'iterations': iters, 'warmup': warm,
            'frac': roof['mfma_kernels']['frac'], 'frac_useful': roof['mfma_kernels']['frac_useful'],
            'frac_note': 'executed (useful) FLOPs of ALL MFMA kernels of the iteration (k_wino_fused + k_igemm / k_gemm_stream + '
                         'k_wgrad) / their hipEvent time / 157.3 TFLOP/s',
            'mfma_kernels_ms': roof['mfma_kernels']['ms_per_step'], 'all_kernels_ms': roof['all_kernels_ms_per_step'],
            'per_class_ms': {k: v['ms_per_step'] for k, v in roof['per_class'].items()}}


def train_workload(args, world, rank, local, dist):
    """Secondary workload (SURVEY 8 row N3 / BASELINE configs[4]): whole training iterations through NativeTrainer."""
    import torch
    import ffrnet_amd
    from ffrnet_amd import synth
    dev = torch.device('cuda', local)
    torch.set_num_threads(max(1, host_cores() // world))
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
    roof = None
    if not args.no_roofline:
        # tr.step() contains the gradient all-reduce, a COLLECTIVE: every rank runs the profiled iterations (a rank that
        # skipped them would leave rank 0 waiting in RCCL forever); only rank 0 instruments its launches and reports.
        roof = train_roofline(eng, lambda: tr.step(non, ocl, label), B, fence, instrument=(rank == 0))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_train()
    if rank == 0:
        print(json.dumps({
            'metric': 'training image pairs/sec (frozen IR-SE50 encoder + RecNet forward/backward + CosFace head + clip + Adam)',
            'value': round(world * B * args.steps / dt, 1), 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[4]: training step, %d pairs per GPU, 112x112x3 fp32' % B,
                       'pairs_per_gpu': B, 'global_pairs': world * B,
                       'parallelism': 'data parallel x%d, one RCCL all-reduce of the flat fp32 gradient buffer per step' % world},
            'roofline': roof, 'cpu_baseline': cpu}))
    if world > 1:
        dist.destroy_process_group()


def main():
    t_start = time.perf_counter()
    args = parse_args()
    # Cross-process GPU memory sharing (RCCL's intra-node transport, torch's CUDA-tensor IPC) needs dmabuf IPC handles on
    # this image's host driver: with the legacy mode RCCL fails in hipIpcGetMemHandle ("invalid argument").  The image
    # exports HSA_ENABLE_IPC_MODE_LEGACY=0 already; a DEFAULT only (a caller's own setting wins), set at the one place
    # every launch mode passes through -- self-launched children inherit it, torch.distributed.run ranks and the
    # single-rank run set it themselves here, before anything initialises the GPU runtime.
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    import ffrnet_amd
    from ffrnet_amd import synth

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
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
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get('FFR_BENCH_DIST_TIMEOUT', '600')))
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local), timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    dev = torch.device('cuda', local)
    if args.workload == 'train':
        return train_workload(args, world, rank, local, dist)

    torch.set_num_threads(max(1, host_cores() // world))     # N ranks synthesise and pack weights side by side
    spec_e, spec_r = state_dict_specs()
    sd_e = synth.synth_state_dict(spec_e)
    sd_r = synth.synth_state_dict(spec_r)
    eng = ffrnet_amd.Engine(local)
    for kv in args.opt:
        eng.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    t_load = time.perf_counter()
    eng.load_encoder(sd_e)
    eng.load_recnet(sd_r)
    t_load = time.perf_counter() - t_load
    strong = args.pairs_per_step > 0
    if strong:
        if (2 * args.pairs_per_step) % world:
            raise SystemExit('--pairs-per-step %d does not split over %d GPUs' % (args.pairs_per_step, world))
        B = 2 * args.pairs_per_step // world
    else:
        B = args.batch
    t_res = time.perf_counter()
    eng.reserve(max(B, 8))           # workspace arena + (first time a batch can use them) the exact-tiling weight sets, derived on the device
    t_res = time.perf_counter() - t_res

    # ---- parity gate: the reference's own outputs for the 8 images of golden G1 (tests/golden/make_golden.py) ----
    g1 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g1_config1.npz'))
    f_new8, f8 = eng.embed(synth.synth_images(8, 112, 112, seed=123).to(dev))
    torch.cuda.synchronize()
    parity = 0.0
    for got, ref in ((f_new8, g1['f_new']), (f8, g1['f'])):
        ref = torch.from_numpy(ref).double()
        parity = max(parity, ((got.double().cpu() - ref).abs().max() / ref.abs().max()).item())
    if not parity < PARITY_TOL:
        print(json.dumps({'error': 'parity check failed', 'parity_checked': parity, 'tolerance': PARITY_TOL}))
        sys.exit(3)

    x = synth.synth_images(B, seed=124 + rank).to(dev)
    # the engine writes f_new and f into the two halves of ONE packed buffer, so the exchange is ONE collective
    # (ffrnet_amd/lfw.py packs the same way): [2][B][512] per rank -> [world][2][B][512]
    pack = torch.empty((2, B, 512), device=dev)
    f_new, f = pack[0], pack[1]
    gathered = torch.empty((world, 2, B, 512), device=dev) if world > 1 else pack.view(1, 2, B, 512)
    coll_ev = []

    def step(timed=False):
        eng.embed(x, out=(f_new, f))
        if world > 1:
            if timed:
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
            dist.all_gather_into_tensor(gathered.view(world * 2 * B, 512), pack.view(2 * B, 512))
            if timed:
                c1.record()
                coll_ev.append((c0, c1))
        # image 2i / 2i+1 of a rank's batch form a verification pair: [world * B/2] pairs per step
        pr = gathered[:, 0].reshape(-1, 2, 512)
        return eng.cosine_scores(pr[:, 0].contiguous(), pr[:, 1].contiguous())

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    startup_s = time.perf_counter() - t_start      # process start -> first step: imports, rendezvous, weight synthesis, packing, parity gate
    for _ in range(args.warmup):
        step()
    fence()
    evs = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        scores = step(timed=True)
        e1.record()
        evs.append((e0, e1))
    fence()
    dt_local = time.perf_counter() - t0
    dt = dt_local
    per_rank = None
    if world > 1:
        # what every rank measured (its own wall clock per step, its step and collective hipEvent medians), so that a
        # scaling loss can be attributed: a slow rank, the collective, or the launch path
        mine = torch.tensor([dt_local / args.steps * 1e3,
                             percentiles([a.elapsed_time(b) for a, b in evs])['median'],
                             percentiles([a.elapsed_time(b) for a, b in coll_ev])['median'], startup_s], device=dev, dtype=torch.float64)
        flat = torch.empty(world * 4, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(flat, mine)
        allr = flat.view(world, 4)
        dt = allr[:, 0].max().item() * args.steps / 1e3                      # MAX over ranks
        per_rank = {'wall_ms_per_step': [round(v, 3) for v in allr[:, 0].tolist()],
                    'step_ms_hipevents_median': [round(v, 3) for v in allr[:, 1].tolist()],
                    'all_gather_ms_hipevents_median': [round(v, 4) for v in allr[:, 2].tolist()],
                    'all_gather_bytes_per_rank': 2 * B * 512 * 4,
                    'startup_s': [round(v, 2) for v in allr[:, 3].tolist()],
                    'startup_note': 'process start -> first step (imports, rendezvous, weight synthesis with host_cores // world torch '
                                    'threads, single-threaded packing, parity gate); the rendezvous / collective timeout is '
                                    'FFR_BENCH_DIST_TIMEOUT (default 600 s)',
                    'note': 'the all-gather event pair also waits for the slowest rank to arrive'}
        # the exchange put every rank's rows where the scoring reads them: this rank's slice is its own output
        assert torch.equal(gathered[rank], pack), 'all-gather: rank %d does not find its own embeddings' % rank
    assert torch.isfinite(scores).all() and scores.numel() == world * B // 2
    step_ms = percentiles([a.elapsed_time(b) for a, b in evs])
    value = world * B * args.steps / dt

    roof = None
    if rank == 0 and not args.no_roofline:
        sha = so_sha256()
        peak_meas, clock = eng.probe_mfma_peak()
        eng.profile_enable(True)
        nprof = 3
        for _ in range(nprof):
            eng.embed(x, out=(f_new, f))
        torch.cuda.synchronize()
        st = eng.profile_read()
        eng.profile_enable(False)
        wf, ig, wn = st['wino_fused'], st['conv_igemm'], st['wino']
        tot_ms = sum(v['ms'] for v in st.values())
        dom = wf if wf['launches'] else ig
        dom_s = dom['ms'] * 1e-3
        dom_tf = dom['flops_executed'] / dom_s / 1e12          # what the matrix cores executed
        dom_useful_tf = dom['flops_useful'] / dom_s / 1e12     # ... without tile / row / channel padding
        mf_ms = wf['ms'] + ig['ms']
        mf_tf = (wf['flops_executed'] + ig['flops_executed']) / (mf_ms * 1e-3) / 1e12
        mf_useful_tf = (wf['flops_useful'] + ig['flops_useful']) / (mf_ms * 1e-3) / 1e12
        hbm = {k: v for k, v in st.items() if k not in ('wino_fused', 'conv_igemm') and v['launches']}
        hbm_ms = sum(v['ms'] for v in hbm.values())
        hbm_bytes = sum(v['bytes'] for v in hbm.values())
        step_s = tot_ms / nprof * 1e-3
        step_exec = sum(v['flops_executed'] for v in st.values()) / nprof
        step_useful = sum(v['flops_useful'] for v in st.values()) / nprof
        compulsory_gb = sum(v['bytes'] for v in st.values()) / nprof / 1e9
        alg_tf = value / world * GFLOP_PER_IMAGE / 1e3          # embeddings/s x 15.1427 GFLOP
        tr = pmc_traffic(sha, batch=B, live=(world == 1))   # child profiler passes only when no other rank waits on this one
        # SURVEY 8(d)'s denominator: ~56 MB of fused activation traffic per image (14.3 GB at batch 256) + 273.2 MB of weights per forward
        survey_gb = 14.3 * B / 256.0 + 0.2732
        ratio_survey = round(tr['hbm_gb_per_step_all_kernels'] / survey_gb, 3) if tr.get('hbm_gb_per_step_all_kernels') else None
        ratio = round(tr['hbm_gb_per_step_all_kernels'] / compulsory_gb, 3) if tr.get('hbm_gb_per_step_all_kernels') else None
        if ratio:
            # VERDICT r03 divided the all-kernel PMC bytes by the compulsory bytes of the HBM-bound classes alone (3.0 x);
            # kept for continuity next to the like-for-like ratio
            tr['ratio_vs_compulsory_all_kernels'] = ratio
            tr['ratio_vs_compulsory_of_hbm_classes_only'] = round(tr['hbm_gb_per_step_all_kernels'] / (hbm_bytes / nprof / 1e9), 3)
            tr['compulsory_definition'] = ('per launch: activations in + out + weights of a convolution / GEMM (logical sizes), '
                                           'input + V for a transform kernel, the operands of an elementwise kernel; summed over '
                                           'ALL launches of a step = %.2f GB (HBM-bound classes alone: %.2f GB)'
                                           % (compulsory_gb, hbm_bytes / nprof / 1e9))
        roof = {'bound': 'mfma',
                'kernel': 'k_wino_fused / k_wino_fused_mixed (the fp32-MFMA GEMMs of every xi + output transform + epilogue of a '
                          'Winograd convolution in one launch: F(4x4,3x3) tiles, on 14x14 maps the exact 4+4+3+3 tiling)',
                'achieved': round(dom_tf, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                # the roofline fraction three ways, all for the dominant kernel over its own hipEvent time:
                'frac': round(dom_tf / PEAK_FP32_MFMA_TFLOPS, 4),                       # FLOPs EXECUTED, padding included
                'frac_useful': round(dom_useful_tf / PEAK_FP32_MFMA_TFLOPS, 4),          # executed minus tile / xi / row / channel padding
                # the round-3-comparable form of the same: the layers' direct-convolution FLOPs / 4 (what an all-F(4x4) tiling without
                # padding would execute) over the same time -- independent of which tile sizes the kernels really use
                'frac_useful_f4x4_equivalent': round(dom['flops'] / 4.0 / dom_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                'frac_algorithmic_survey_8d': round(alg_tf / PEAK_FP32_MFMA_TFLOPS, 4),  # whole step, direct-convolution count
                'traffic': tr.get('hbm_bytes_per_launch'),              # HBM bytes per launch of the dominant kernel (PMC), or null
                # all kernels: PMC bytes per step / SURVEY 8(d)'s algorithmic bytes (56 MB per image + the weights once): the wasted-re-read figure
                'traffic_ratio_vs_compulsory': ratio_survey,
                'traffic_compulsory_gb_survey_8d': round(survey_gb, 3),
                'traffic_compulsory_definition': 'SURVEY 8(d): 14.3 GB x batch / 256 of activations (56 MB per image) + 0.2732 GB of weights per forward; '
                                                 'valid for THIS workload only (112x112 IR-SE50 + RecNet embed, default options); round 4 and earlier '
                                                 'printed the per-launch-operand ratio under the key traffic_ratio_vs_compulsory (now '
                                                 'traffic_ratio_vs_per_launch_operands)',
                # ... / the sum of every launch's own operands (a V tensor counts for the kernel that writes it AND the one that reads it)
                'traffic_ratio_vs_per_launch_operands': ratio,
                'note': 'frac: FLOPs the matrix cores EXECUTED in the fused Winograd launches (2*xi*ceil(T/32)*32*cin_pad*cout_pad per '
                        'launch and tile type, xi = 36, or 36/32/32/28 for the four tile types of a 14x14 map) / their hipEvent '
                        'time / peak.  frac_useful: the same without padding (tiles hanging over 7x7 maps, padded xi, rows beyond T, '
                        'zero-padded channels).  Executing FEWER FLOPs for the same result (the exact tiling of 14x14 maps, round '
                        '4) lowers frac and raises frac_useful and the throughput.  '
                        'frac_algorithmic_survey_8d: embeddings/s x 15.1427 GFLOP (SURVEY 8d counts every convolution as a '
                        'direct one) / peak for the WHOLE step; it exceeds 1 because Winograd F(4x4,3x3) executes 36 instead of '
                        '144 multiplies per 4x4 output tile and channel pair (results verified in this process: parity_checked)',
                'launches_per_step': dom['launches'] // nprof,
                'avg_launch_us': round(dom['ms'] * 1e3 / max(1, dom['launches']), 2),
                'gflop_executed_per_launch': round(dom['flops_executed'] / max(1, dom['launches']) / 1e9, 3),
                'gflop_useful_per_launch': round(dom['flops_useful'] / max(1, dom['launches']) / 1e9, 3),
                'compulsory_gb_per_step': round(compulsory_gb, 3),
                'traffic_detail': tr,
                # what the kernel is co-limited by: operand fragments streamed L2 -> registers; a 32x64 tile
                # per xi and 8-channel chunk moves (32 + 64) * 8 * 4 bytes for 2 * 32 * 64 * 8 flops
                'operand_stream': {'bytes_per_flop': 0.09375,
                                   'tb_per_s': round(dom_tf * 0.09375, 2),
                                   'bytes_per_clk_per_cu': round(dom_tf * 1e12 * 0.09375 / (256 * clock * 1e9), 1),
                                   'note': 'V and U fragments of k_wino_fused, L2 (U) / Infinity Cache or L2 (V) -> VGPRs; '
                                           'averaged over the launch incl. its transform and epilogue phases (K loop alone: 21 B/clk/CU)'},
                'peak_measured': {'tflops': round(peak_meas, 1), 'shader_clock_ghz': round(clock, 3),
                                  'frac_of_measured': round(dom_tf / peak_meas, 4),
                                  'how': 'ffr_probe_mfma_peak: register-resident v_mfma_f32_32x32x2_f32 loop on every CU of '
                                         'this device, hipEvent time, clock from s_memtime / s_memrealtime'},
                'per_bound': {
                    'mfma': {'kernels': 'k_wino_fused + k_igemm (stride-2 convs, 1x1 shortcuts, FC)',
                             'executed_tflops': round(mf_tf, 2), 'frac': round(mf_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                             'frac_useful': round(mf_useful_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                             'ms_per_step': round(mf_ms / nprof, 3)},
                    'hbm': {'kernels': 'k_wino_in_c, k_combine, k_se_*, k_stem, RecNet operators, layout',
                            'compulsory_gb_per_step': round(hbm_bytes / nprof / 1e9, 3),
                            'achieved_tbs': round(hbm_bytes / (hbm_ms * 1e-3) / 1e12, 3) if hbm_ms else None,
                            'frac': round(hbm_bytes / (hbm_ms * 1e-3) / 1e12 / PEAK_HBM_TBS, 4) if hbm_ms else None,
                            'ms_per_step': round(hbm_ms / nprof, 3)},
                    'whole_step': {'executed_tflops': round(step_exec / step_s / 1e12, 2),
                                   'frac_of_mfma_peak': round(step_exec / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                   'frac_useful': round(step_useful / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
                    'note': 'ms_per_step figures are sums of per-launch hipEvent pairs: each pair also contains the dispatch gap '
                            'in front of its launch, so the classes add up to all_kernels_ms_per_step, which is slightly MORE '
                            'than the back-to-back ms_per_step of the timed region'},
                'effective_tflops_algorithmic': round(alg_tf, 2),
                'kernel_ms_per_step': {k: round(v['ms'] / nprof, 3) for k, v in st.items() if v['launches']},
                'all_kernels_ms_per_step': round(tot_ms / nprof, 3),
                'so_sha256': sha}

    # load / pack seconds and device bytes of the handle: wall clock of ffr_load_encoder + ffr_load_recnet (host-side
    # BN folding, Winograd weight transforms, fragment orders; torch.set_num_threads above does not matter: the packer is one
    # thread) and of the first ffr_reserve, and what the handle holds afterwards
    ms_ = eng.memory_stats()
    load = {'startup_s': round(startup_s, 2), 'load_s': round(t_load, 3), 'reserve_s': round(t_res, 3),
            'encoder_load_s': round(ms_['encoder_load_seconds'], 3), 'recnet_load_s': round(ms_['recnet_load_seconds'], 3),
            'mixed_tile_pack_s': round(ms_['mixed_tile_pack_seconds'], 4),
            'weight_gb': round((ms_['encoder_weight_bytes'] + ms_['recnet_weight_bytes']) / 1e9, 3),
            'mixed_tile_weight_gb': round(ms_['mixed_tile_weight_bytes'] / 1e9, 3),
            'workspace_gb': round(ms_['workspace_bytes'] / 1e9, 3),
            'note': 'mixed_tile_*: the three extra Winograd weight sets of the exact 14x14 tiling, derived on the device at the first '
                    'reserve / forward of a batch that uses them (>= 256 images; >= 128 for the 256 -> 512 layer), 0 for handles that only see smaller batches'}

    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary and not strong:
        # configs[1]: backbone only (featmap + f); and the 112x96 of the metric label, which exists for the trunk only
        ms1 = percentiles(timed_events(lambda: eng.encoder_forward(x), 3, 10))
        x96 = synth.synth_images(B, 112, 96, seed=125).to(dev)
        ms96 = percentiles(timed_events(lambda: eng.encoder_forward(x96, want_f=False), 3, 10))
        secondary = [
            {'workload': 'configs[1]: IR-SE50 backbone only, batch %d, 112x112 -> featmap [512,7,7] + f [512]' % B,
             'value': round(B / ms1['median'] * 1e3, 1), 'unit': 'images/s', 'ms': ms1,
             'effective_tflops_algorithmic': round(B * 12.5934 / ms1['median'], 2)},
            {'workload': 'TRUNK ONLY at 112x96 (the resolution of the metric label; the reference cannot embed it: '
                         'lfw/gen_lfw112x96.py:16 vs model_ir_se50.py:124), batch %d -> featmap [512,7,6]' % B,
             'value': round(B / ms96['median'] * 1e3, 1), 'unit': 'images/s', 'ms': ms96,
             'effective_tflops_algorithmic': round(B * GFLOP_TRUNK_96 / ms96['median'], 2)}]
        secondary.append(lfw_protocol_line(eng, dev))
        secondary.append(train_step_line(eng, sd_r, dev))
    if world > 1 and not strong and not args.no_secondary and args.strong_pairs > 0:
        # the default N-GPU line is weak scaling (256 images per GPU); north_star's ">= 6x at 8 GPUs" is about configs[3], whose
        # per-GPU batch SHRINKS with N: measured here in the same run, on every rank
        entry = strong_scaling_line(eng, dist, world, rank, dev, min(args.steps, 20), min(args.warmup, 5), args.strong_pairs)
        if rank == 0 and entry:
            secondary = [entry]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd_e, sd_r)

    if rank == 0:
        wl = ('configs[3] shape: %d verification pairs per step split over %d GPU(s) = %d images per GPU, '
              % (args.pairs_per_step, world, B)) if strong else \
             ('configs[2]: IR-SE50 + RecBlock (spatial+channel) forward, batch %d per GPU, ' % B)
        out = {'metric': 'face embeddings/sec (IR-SE50 + RecNet forward, 112x112 fp32) at batch 256 per GPU' if not strong
               else 'face embeddings/sec (IR-SE50 + RecNet forward, 112x112 fp32), %d pairs per step' % args.pairs_per_step,
               'value': round(value, 1), 'unit': 'embeddings/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
               'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f32',
               'data': 'synthetic',
               'config': {'workload': wl + '112x112x3 fp32 -> f_new,f [512]; + all-gather of embeddings and pair '
                                         'cosine scores (112x96 of the BASELINE label cannot be embedded by the '
                                         'reference: BASELINE.md)',
                          'batch_per_gpu': B, 'global_batch': world * B, 'gflop_per_image': GFLOP_PER_IMAGE,
                          'parallelism': 'image-sharded x%d, RCCL all-gather of embeddings' % world},
               'step_ms_hipevents': step_ms, 'per_rank': per_rank, 'options': args.opt or None,
               'parity_checked': {'max_rel_err_vs_reference_golden_G1': parity, 'tolerance': PARITY_TOL},
               'load': load, 'roofline': roof, 'cpu_baseline': cpu, 'secondary': secondary}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
