"""RCCL on the hardware a 1-GPU box has: ONE rank with the `nccl` backend (= RCCL on ROCm).  The collectives are the
product's own calls on the product's own buffers -- the all-gather of the embeddings (bench.py / lfw.py), the bucketed
gradient all-reduce on the second stream behind the per-bucket events (train.py: average_gradients_overlapped, on the
zero-copy views of the native gradient buffer), the parameter broadcast -- so RCCL's argument checks (dtype,
contiguity, device pointers that torch did not allocate) and the stream/event hand-over run for real; with one rank
every collective is the identity, which is what is asserted."""
import datetime, json, os, sys
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd  # noqa: E402
from ffrnet_amd import synth, train as ftrain  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29561')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0),
                        timeout=datetime.timedelta(seconds=120))
specs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g0_state_dict_keys.json')))
eng = ffrnet_amd.Engine(0)
eng.load_encoder(synth.synth_state_dict(specs['encoder']))
eng.load_recnet(synth.synth_state_dict(specs['recnet']))

# 1. embeddings: all-gather into the scoring buffers, as bench.py does per step
x = synth.synth_images(16, seed=3).cuda()
f_new, f = eng.embed(x)
g_new, g_old = torch.empty_like(f_new), torch.empty_like(f)
dist.all_gather_into_tensor(g_new, f_new)
dist.all_gather_into_tensor(g_old, f)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(g_new, f_new) and torch.equal(g_old, f)

# 2. training: blocking exchange vs the bucketed exchange on the second stream, both through RCCL
non, ocl, label = (t.cuda() for t in synth.synth_train_batch(4, seed=700))
res = []
for overlapped in (False, True):
    tr = ffrnet_amd.NativeTrainer(eng, synth.synth_state_dict(specs['recnet']), lr=1e-3)
    dist.broadcast(tr.flat_params, 0)
    eng.train_iteration(non, ocl, label, tr.loss_weight)
    if overlapped:
        comm = torch.cuda.Stream(device=0)
        ftrain.average_gradients_overlapped(eng, tr.flat_grads, comm)
    else:
        dist.all_reduce(tr.flat_grads)
        tr.flat_grads.div_(1)
    eng.train_adam_step(tr.lr, tr.betas, 1e-8, tr.weight_decay, tr.clip_value)
    torch.cuda.synchronize()
    res.append(tr.flat_params.clone())
assert torch.isfinite(res[0]).all() and torch.equal(res[0], res[1]), 'bucketed exchange differs from the blocking one'
# and against the same iteration with no process group in the way
tr = ffrnet_amd.NativeTrainer(eng, synth.synth_state_dict(specs['recnet']), lr=1e-3, overlap=False)
eng.train_iteration(non, ocl, label, tr.loss_weight)
eng.train_adam_step(tr.lr, tr.betas, 1e-8, tr.weight_decay, tr.clip_value)
torch.cuda.synchronize()
assert torch.equal(tr.flat_params, res[0]), 'RCCL identity all-reduce changed the gradients'
dist.destroy_process_group()
print('backend', 'nccl (RCCL)', 'buckets', [b for b, _, _ in eng.train_buckets()])
print('OK')
