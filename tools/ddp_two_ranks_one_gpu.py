"""Two training processes on ONE GPU with the gloo backend (RCCL refuses two ranks on one device): after an
iteration on different batches both ranks must hold bitwise identical parameters -- the flat gradient buffer,
a zero-copy torch view of native device memory, really went through the all-reduce."""
import json, os, sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q):
    import ffrnet_amd
    from ffrnet_amd import synth
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    specs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g0_state_dict_keys.json')))
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(synth.synth_state_dict(specs['encoder']))
    non, ocl, label = (t.cuda() for t in synth.synth_train_batch(4, seed=700 + rank))
    # one blocking all-reduce of the whole flat buffer after the backward ...
    tr = ffrnet_amd.NativeTrainer(eng, synth.synth_state_dict(specs['recnet']), lr=1e-3, overlap=False)
    tr.broadcast_params(0)
    tr.step(non, ocl, label)
    torch.cuda.synchronize()
    p_blocking = tr.flat_params.clone()
    # ... and the bucketed exchange on a second stream under the backward must give the same parameters
    tr = ffrnet_amd.NativeTrainer(eng, synth.synth_state_dict(specs['recnet']), lr=1e-3, overlap=True)
    tr.broadcast_params(0)
    items = tr.step(non, ocl, label)
    torch.cuda.synchronize()
    assert torch.equal(tr.flat_params, p_blocking), 'overlapped exchange differs from the blocking one'
    buckets = eng.train_buckets()
    assert sum(c for _, _, c in buckets) == tr.flat_grads.numel() and [b for b, _, _ in buckets] == [4, 2, 1, 3, 0]
    p = tr.flat_params.cpu()
    q.put((rank, [float(x) for x in items], float(p.double().sum()), float(p.double().abs().sum()), p[::100003].tolist()))
    dist.destroy_process_group()


if __name__ == '__main__':
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29533, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(60)
    finally:
        for p in procs:         # a rank that failed must not leave its peer waiting in a collective
            if p.is_alive():
                p.terminate()
    print('losses differ (different batches):', res[0][1] != res[1][1])
    print('parameters identical after the step:', res[0][2:] == res[1][2:])
    assert res[0][1] != res[1][1] and res[0][2:] == res[1][2:]
    print('OK')
