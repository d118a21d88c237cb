"""CPU-only checks of the host side: the state_dict contract of the shells, the C-ABI
library's exported symbols (no compute without a GPU), loud failure without the native
path, and the sharded verification harness on 2 gloo ranks."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import ffrnet_amd
from ffrnet_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_key_contract(specs):
    """402 / 121 entries, same names, order and shapes as the reference modules (G0)."""
    enc = ffrnet_amd.Backbone(num_layers=50, drop_ratio=0.6, mode='ir_se')
    rec = ffrnet_amd.RecNet(channel=512, shape=7, norm_type='bn', relu_type='prelu')
    for mod, key in ((enc, 'encoder'), (rec, 'recnet')):
        sd = mod.state_dict()
        assert list(sd.keys()) == list(specs[key].keys())
        for k, v in sd.items():
            assert list(v.shape) == specs[key][k], k
    assert len(specs['encoder']) == 402 and len(specs['recnet']) == 121
    # strict load of a reference-shaped state_dict, clone flow of models/trainer.py:98-113
    sd_e = ffrnet_amd.synth.synth_state_dict(specs['encoder'])
    enc.load_state_dict(sd_e)
    enc2 = ffrnet_amd.Backbone(50, 0.6, 'ir_se')
    enc2.load_state_dict(enc.state_dict())
    assert torch.equal(enc2.state_dict()['body.3.shortcut_layer.0.weight'],
                       sd_e['body.3.shortcut_layer.0.weight'])


def test_unsupported_constructor_arguments():
    assert len(ffrnet_amd.Backbone(100, 0.6, 'ir').body) == 49 and len(ffrnet_amd.Backbone(152, 0.6, 'ir_se').body) == 50
    with pytest.raises(NotImplementedError):
        ffrnet_amd.RecNet(norm_type='in')
    with pytest.raises(AssertionError):
        ffrnet_amd.Backbone(34, 0.6, 'ir_se')


def test_no_cpu_fallback():
    """The product fails loudly off-GPU: no oracle, no stock-torch path."""
    enc = ffrnet_amd.Backbone(50, 0.6, 'ir_se').eval()
    rec = ffrnet_amd.RecNet().eval()
    with pytest.raises(RuntimeError):
        enc(torch.zeros(1, 3, 112, 112))
    with pytest.raises(RuntimeError):
        rec(torch.zeros(1, 512, 7, 7))
    with pytest.raises(NotImplementedError):
        rec(torch.zeros(1, 512, 7, 7), label=torch.zeros(1))          # label branch needs train()
    with pytest.raises(RuntimeError):
        rec.train()(torch.zeros(2, 512, 7, 7), label=torch.zeros(2))   # train branch: device tensors only
    src = ''
    for fn in os.listdir(os.path.join(ROOT, 'ffr-net_amd')):
        if fn.endswith('.py'):
            src += open(os.path.join(ROOT, 'ffr-net_amd', fn)).read()
    assert not re.search(r'^\s*(import|from)\s+\.*(oracle|ffr_oracle)', src, re.M)


def test_library_reads_no_environment():
    """kernel selection must not depend on environment variables -- every knob is an option of the handle
    (ffr_set_option).  Static check of the native sources and of the built library's dynamic imports."""
    src = ''
    for fn in os.listdir(os.path.join(ROOT, 'ffr-net_amd', 'csrc')):
        src += open(os.path.join(ROOT, 'ffr-net_amd', 'csrc', fn)).read()
    assert 'getenv' not in src and 'environ[' not in src and 'secure_getenv' not in src
    out = subprocess.run(['nm', '-D', '--undefined-only', native.lib_path()], capture_output=True, text=True)
    if out.returncode == 0:
        assert 'getenv' not in out.stdout


def test_c_abi_exports_every_declared_symbol():
    """The .so loads without a GPU and exports each function include/*.h declares."""
    path = native.lib_path()
    assert os.path.exists(path), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(path)
    import glob
    hdr = ''.join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, 'include', '*.h'))))
    declared = set(re.findall(r'\b(ffr_[a-z_0-9]+)\s*\(', hdr))
    bound = {n for n, _, _ in native.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        getattr(lib, name)
    lib.ffr_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.ffr_version()
    if not torch.cuda.is_available():
        h = ctypes.c_void_p(0)
        rc = lib.ffr_create(ctypes.byref(h), 0)
        assert rc != 0 and not h.value          # no device -> error code, never a CPU path


def test_integration_lists_the_sources():
    """INTEGRATION.md section 5 names the translation units of the library; held to SOURCES of
    __graft_entry__.py (the experiment kernel k_wino_fused_q left the product in round 5 and must not come back unnoticed),
    and the built library exports no symbol of it."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'csrc/\{([^}]*)\}', text)
    assert m, 'INTEGRATION.md section 5 lost its source list'
    assert sorted(m.group(1).split(',')) == sorted(g.SOURCES)
    assert all(os.path.exists(os.path.join(g.CSRC, f)) for f in g.SOURCES)
    assert 'wino_fused_q.hip' not in g.SOURCES
    out = subprocess.run(['nm', '-D', native.lib_path()], capture_output=True, text=True)
    if out.returncode == 0:
        assert 'wino_fused_q' not in out.stdout
    raw = open(native.lib_path(), 'rb').read()
    assert b'k_wino_fused_q' not in raw           # no such kernel in the embedded code object either


def test_committed_profiles_agree_with_their_bench_lines():
    """The round's committed measurement set is self-consistent: tools/roofline_check.py recomputes the roofline block of
    profiles/r06_bench.json from the rocprofv3 summary next to it (launches, average launch time, executed / useful FLOPs from a layer
    table, the three fractions) and the training classes of profiles/r06_train_step.json from theirs; and the PMC summary bench.py
    would quote carries the hash of the same build as the bench line."""
    import json
    prof = os.path.join(ROOT, 'profiles')
    for extra in ([], ['--train']):
        stats = 'r06_train_step_kernel_stats.csv' if extra else 'r06_bench_batch256_kernel_stats.csv'
        line = 'r06_train_step.json' if extra else 'r06_bench.json'
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'roofline_check.py')] + extra +
                             [os.path.join(prof, stats), os.path.join(prof, line)], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
    bench = json.loads([l for l in open(os.path.join(prof, 'r06_bench.json')) if l.startswith('{')][-1])
    pmc = json.load(open(os.path.join(prof, 'r06_pmc_hbm_traffic.json')))
    assert pmc['so_sha256'] == bench['roofline']['so_sha256'] and pmc['batch'] == 256
    assert abs(bench['roofline']['traffic_ratio_vs_compulsory'] - pmc['gb_per_step'] / (14.3 + 0.2732)) < 0.01
    assert bench['secondary'][-1]['unit'] == 'pairs/s' and bench['secondary'][-1]['ms_per_iteration'] > 0       # the training line


def test_design_documents_and_option_table_stay_in_shape():
    """DESIGN.md is the CURRENT design (<= 300 lines), the lab notebook lives in EXPERIMENTS.md and its index
    names every profiles/r0N_exp_* artefact exactly once; the handle has <= 18 options and include/ffrnet.h documents each of them
    (and none that is gone)."""
    design = open(os.path.join(ROOT, 'DESIGN.md')).read()
    assert design.count('\n') <= 300
    exp = open(os.path.join(ROOT, 'EXPERIMENTS.md')).read()
    index = exp[exp.index('## Index of experiment artefacts'):exp.index('## Round 6')]
    files = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if re.match(r'r0\d_exp_', f))
    assert files, 'no experiment artefacts found'
    for f in files:
        assert index.count('`%s`' % f) == 1, f
    assert len(re.findall(r'^\| `r0\d_exp_', index, re.M)) == len(files)          # and no row for a file that does not exist
    eng = open(os.path.join(ROOT, 'ffr-net_amd', 'csrc', 'engine.cpp')).read()
    table = eng[eng.index('const OptEntry OPTIONS[] = {'):]
    table = table[:table.index('};')]
    names = re.findall(r'\{"(\w+)"', table)
    assert 10 <= len(names) <= 18 and len(set(names)) == len(names)
    header = open(os.path.join(ROOT, 'include', 'ffrnet.h')).read()
    doc = header[header.index('Experiment knobs of one handle'):header.index('int ffr_set_option')]
    for n in names:
        assert '"%s"' % n in doc, n
    for gone in ('wino_112', 'wf_halfblocks', 'wf_mapv', 'wf_mapx', 'wf_maph', 'wm_xcdpairs', 'wino_slice_mb', 'gs_tile', 'wino_oi',
                 'se_fuse', 'igemm_tile64'):
        assert '"%s"' % gone not in doc and gone not in names
    assert 'FFR_WF_NPRE' not in open(os.path.join(ROOT, 'ffr-net_amd', 'csrc', 'wino_fused.hip')).read()


def test_weight_cache_sees_submodule_surgery():
    """replacing a Parameter / buffer on a SUB-module must invalidate the packed native copy.  The
    signature the shells compare on every forward is (identity, version) of each tensor in the tree."""
    rec = ffrnet_amd.RecNet()
    dev = torch.device('cuda', 0)
    sig0, keep0 = rec._signature(dev)
    assert len(keep0) == 121 and rec._signature(dev)[0] == sig0
    rec.Conv4Space[0].conv2d.weight = torch.nn.Parameter(torch.zeros_like(rec.Conv4Space[0].conv2d.weight))
    sig1, _ = rec._signature(dev)
    assert sig1 != sig0
    with torch.no_grad():
        rec.Conv4Merge[0].norm.norm.running_mean.add_(1.0)              # in-place edit: version counter
    sig2, _ = rec._signature(dev)
    assert sig2 != sig1
    sub = rec.ChannelFlipMerge[0].norm.norm
    sub.load_state_dict({k: v.clone() for k, v in sub.state_dict().items()}, assign=True)
    sig3, _ = rec._signature(dev)
    assert sig3 != sig2
    rec.load_state_dict(rec.state_dict())                                # copy_ into every tensor
    assert rec._signature(dev)[0] != sig3
    enc = ffrnet_amd.Backbone(50, 0.6, 'ir_se')
    assert len(enc._tensors()) == 402


def test_shard_bounds_cover_everything():
    for n in (1, 7, 512, 513):
        for world in (1, 2, 8):
            got = []
            for r in range(world):
                lo, hi = ffrnet_amd.lfw.shard_bounds(n, r, world)
                got += list(range(lo, hi))
            assert got == list(range(n))


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import ffrnet_amd
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[3]
dist.init_process_group('gloo', rank=rank, world_size=world)
P = torch.randn(3 * 112 * 112, 512, generator=torch.Generator().manual_seed(5)) / 100
def embed(img):                      # stand-in embedder (host-logic test, CPU)
    e = img.reshape(img.size(0), -1) @ P
    return e, torch.tanh(e)
calls = []
def counting(img):
    calls.append(img.size(0)); return embed(img)
i1, i2, lab = ffrnet_amd.synth.synth_pairs(21, seed=3, block=6)
loader = [dict(img1=i1[s:s+8], img2=i2[s:s+8], label=lab[s:s+8], idx=torch.arange(s, min(s+8, 21)))
          for s in (0, 8, 16)]
pn, p = ffrnet_amd.lfw.calculate_distance(loader, counting)
np.save(sys.argv[4] + '.%%d.npy' %% rank, np.concatenate([pn, p], 1))
print('calls', rank, calls)
st = ffrnet_amd.lfw.last_feed_stats
print('feed', rank, st['batches'], st['shard_bytes'], st['full_batch_bytes'], st['h2d_bytes'])
# the reference-shaped entry point lfw_eval.get_avg_accuracy(encoder, recnet, data_loader) with two foreign
# modules (called in the reference's order), default scoring / fold protocol, sharded over the ranks
class Enc(torch.nn.Module):
    def forward(self, x):
        e = x.reshape(x.size(0), -1) @ P
        return e.reshape(-1, 512, 1, 1), torch.tanh(e)
class Rec(torch.nn.Module):
    def forward(self, fm, label=None):
        return fm.reshape(-1, 512), fm
acc_new, acc, det = ffrnet_amd.lfw.get_avg_accuracy(Enc(), Rec(), loader, n_folds=3, details=True)
print('acc', rank, repr(acc_new), repr(acc), float(np.abs(det['pred_new'] - pn).max()))
dist.destroy_process_group()
'''


def test_sharded_verification_two_ranks_gloo(tmp_path):
    """world_size 2 over gloo: every rank embeds only its shard, the all-gathered result
    equals the single-process result."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % {'root': ROOT})
    port = str(29500 + os.getpid() % 2000)
    out = str(tmp_path / 'res')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), '2', port, out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    r0, r1 = np.load(out + '.0.npy'), np.load(out + '.1.npy')
    assert np.array_equal(r0, r1)
    # single process reference
    P = torch.randn(3 * 112 * 112, 512, generator=torch.Generator().manual_seed(5)) / 100

    def embed(img):
        e = img.reshape(img.size(0), -1) @ P
        return e, torch.tanh(e)
    i1, i2, lab = ffrnet_amd.synth.synth_pairs(21, seed=3, block=6)
    loader = [dict(img1=i1[s:s + 8], img2=i2[s:s + 8], label=lab[s:s + 8],
                   idx=torch.arange(s, min(s + 8, 21))) for s in (0, 8, 16)]
    pn, p = ffrnet_amd.lfw.calculate_distance(loader, embed)
    assert np.abs(np.concatenate([pn, p], 1) - r0).max() < 1e-5
    assert 'calls 0 [8, 8, 6]' in logs[0] and 'calls 1 [8, 8, 4]' in logs[1]
    # the input side: a rank slices its shard on the host before anything else happens to the images, so the bytes it
    # handles are its shard's (2 images per pair x 4 B x 3*112*112), never the whole pair batch's
    img = 3 * 112 * 112 * 4
    for r, pairs in ((0, 4 + 4 + 3), (1, 4 + 4 + 2)):
        line = [ln for ln in logs[r].splitlines() if ln.startswith('feed %d' % r)][0].split()
        assert [int(v) for v in line[2:]] == [3, 2 * pairs * img, 2 * 21 * img, 0], line
    # get_avg_accuracy(encoder, recnet, loader): same numbers on both ranks and as the single-process protocol
    a_new = ffrnet_amd.lfw.get_accuracy_from_predicts(pn, 3)[0]
    a_old = ffrnet_amd.lfw.get_accuracy_from_predicts(p, 3)[0]
    for r in range(2):
        line = [ln for ln in logs[r].splitlines() if ln.startswith('acc %d' % r)][0].split()
        assert abs(float(line[2]) - a_new) < 1e-12 and abs(float(line[3]) - a_old) < 1e-12 and float(line[4]) < 1e-5
    assert 0.0 < a_new <= 1.0


def test_checkpoint_containers_round_trip(tmp_path, specs):
    """the reference's two checkpoint containers (plain se50.pth, gzip
    {'RecNet','optimizer','epoch','iter'}) load into the shells with the reference's semantics."""
    from ffrnet_amd import checkpoint as ck
    sd_e = ffrnet_amd.synth.synth_state_dict(specs['encoder'])
    sd_r = ffrnet_amd.synth.synth_state_dict(specs['recnet'])
    enc_path = str(tmp_path / 'se50.pth')
    torch.save(sd_e, enc_path)
    enc = ffrnet_amd.ir_se_50_512(enc_path)                       # model_ir_se50.py:143-154
    assert torch.equal(enc.state_dict()['output_layer.3.weight'], sd_e['output_layer.3.weight'])
    rec = ffrnet_amd.RecNet()
    rec.load_state_dict(sd_r)
    os.makedirs(tmp_path / 'ckpt')
    ck.save_recnet_checkpoint(rec, str(tmp_path / 'ckpt' / '0000400.pth.gzip'), extra_info={'epoch': 3, 'iter': 400})
    ck.save_recnet_checkpoint(rec, str(tmp_path / 'ckpt' / 'latest.pth.gzip'), extra_info={'epoch': 4, 'iter': 555})
    assert ck.latest_checkpoint(str(tmp_path / 'ckpt')).endswith('latest.pth.gzip')
    rec2 = ffrnet_amd.RecNet()
    pos = ck.load_recnet_checkpoint(rec2, ck.latest_checkpoint(str(tmp_path / 'ckpt')))
    assert pos == {'epoch': 4, 'iter': 555}
    for k, v in rec.state_dict().items():
        assert torch.equal(v, rec2.state_dict()[k]), k
    # strict=False like the reference: a checkpoint without the classifier still loads
    w = ck.load(str(tmp_path / 'ckpt' / '0000400.pth.gzip'))
    del w['RecNet']['classifier.weight']
    ck.save(w, str(tmp_path / 'noclf.pth.gzip'))
    ck.load_recnet_checkpoint(ffrnet_amd.RecNet(), str(tmp_path / 'noclf.pth.gzip'))


def _avg_grad_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ffrnet_amd import train
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    train.average_gradients(flat)
    q.put((rank, flat[:4].tolist(), float(flat.sum())))
    dist.destroy_process_group()


def test_gradient_averaging_two_ranks():
    """The data-parallel exchange of the training step: one all-reduce of the flat gradient buffer, mean over ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_avg_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, head, total in res:
        assert head == [0.0, 1.5, 3.0, 4.5]
        assert abs(total - 1.5 * 999 * 1000 / 2) < 1e-3


class _FakeDeviceTensor(object):
    """Stands in for a tensor on a GPU this container does not have: _check_dev only looks at type, device, dtype, shape."""

    def __new__(cls, index, shape=(2, 3, 112, 112)):
        from unittest import mock
        t = mock.Mock(spec=torch.Tensor)
        t.is_cuda, t.device, t.dtype, t.shape = True, torch.device('cuda', index), torch.float32, torch.Size(shape)
        return t


def test_boundary_rejects_tensors_of_another_device():
    """a tensor that lives on GPU 0 handed to the handle of GPU 3 must raise, not run."""
    here, there = _FakeDeviceTensor(3), _FakeDeviceTensor(0)
    native._check_dev(here, 'x', (3, 112, 112), device=torch.device('cuda', 3))
    with pytest.raises(RuntimeError, match=r'cuda:0 but this Engine lives on cuda:3'):
        native._check_dev(there, 'x', (3, 112, 112), device=torch.device('cuda', 3))
    # every Engine method that takes a device tensor passes its own device to the check
    src = open(os.path.join(ROOT, 'ffr-net_amd', 'native.py')).read()
    body = src.split('class Engine(object):', 1)[1]
    calls = re.findall(r'_check_dev\((.*)\)', body)
    assert len(calls) >= 18 and all('device=self.device' in c or 'device=self.engine.device' in c for c in calls), calls
    # out= buffers of Engine.embed get the same check (no GPU: run the method body against a stub handle)
    eng = native.Engine.__new__(native.Engine)
    eng.device = torch.device('cuda', 3)
    x = _FakeDeviceTensor(3)
    x.contiguous.return_value, x.size.return_value = x, 2
    with pytest.raises(RuntimeError, match=r'out\[0\] is on cuda:0'):
        eng.embed(x, out=(_FakeDeviceTensor(0, (2, 512)), _FakeDeviceTensor(3, (2, 512))))
    eng._h = None       # __del__ -> close() finds nothing to destroy


def test_shells_refuse_data_parallel_replicas():
    """models/trainer.py:70-72 wraps every call in nn.parallel.data_parallel; with more than one gpu id torch replicates
    the module per device in threads.  The shells state the one-process-per-GPU rule instead of silently re-packing
    all weights per replica and forward."""
    for shell, arg in ((ffrnet_amd.Backbone(50, 0.6, 'ir_se'), torch.zeros(1, 3, 112, 112)),
                       (ffrnet_amd.RecNet(), torch.zeros(1, 512, 7, 7))):
        shell.eval()
        rep = shell._replicate_for_data_parallel()
        with pytest.raises(RuntimeError, match='ONE PROCESS PER GPU'):
            rep(arg)
    rep = ffrnet_amd.RecNet().train()._replicate_for_data_parallel()
    with pytest.raises(RuntimeError, match='ONE PROCESS PER GPU'):
        rep(torch.zeros(1, 512, 7, 7), torch.zeros(1, dtype=torch.long))
    # the original (non-replica) module still fails for the documented reason only: no CPU path
    with pytest.raises(RuntimeError, match='no CPU path'):
        ffrnet_amd.RecNet().eval()(torch.zeros(1, 512, 7, 7))


def test_shard_feeder_slices_before_it_copies():
    """ShardFeeder on host tensors without a device: every rank is handed exactly its contiguous shard, in loader order,
    ragged last batch and empty shards included."""
    i1 = torch.arange(21 * 3 * 4 * 4, dtype=torch.float32).reshape(21, 3, 4, 4)
    i2 = -i1
    loader = [dict(img1=i1[s:s + 8], img2=i2[s:s + 8], label=torch.zeros(8), idx=torch.arange(8)) for s in (0, 8, 16)]
    world = 4
    seen = [[] for _ in range(world)]
    for r in range(world):
        fd = ffrnet_amd.lfw.ShardFeeder(loader, r, world, device=None)
        for (data, both, m, n), s0 in zip(fd, (0, 8, 16)):
            lo, hi = ffrnet_amd.lfw.shard_bounds(n, r, world)
            assert m == hi - lo
            if m:
                assert torch.equal(both[:m], i1[s0 + lo:s0 + hi]) and torch.equal(both[m:], i2[s0 + lo:s0 + hi])
                seen[r].append((s0 + lo, s0 + hi))
            else:
                assert both is None
        assert fd.stats['batches'] == 3 and fd.stats['h2d_bytes'] == 0
        assert fd.stats['shard_bytes'] == 2 * sum(b - a for a, b in seen[r]) * 3 * 4 * 4 * 4
    assert sorted(x for r in seen for a, b in r for x in range(a, b)) == list(range(21))
