import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import ffrnet_amd
eng = ffrnet_amd.Engine(0); eng.reserve(256)
N, cin, cout = 256, 25088, 512
x = torch.randn(N, 1, 1, cin, device='cuda'); w = torch.randn(cout, cin, device='cuda') * 0.01
bias = torch.zeros(cout, device='cuda'); out = torch.empty(N, 1, 1, cout, device='cuda')
for t in (0, 1, 2, 3, 4):
    kw = dict(x=x, N=N, H=1, W=1, in_pitch=cin, cin_pad=cin, w=w, bias=bias, slope=None, resid=None, res_pitch=0, out=out,
              out_pitch=cout, out_coff=0, cout_store=cout, cout_pad=cout, R=1, S=1, stride=1, pad=0, pad_mode=0, border_bias=0,
              flags=0, tile=t, splitk=0)
    for _ in range(3): eng.op_conv(**kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.op_conv(**kw)
    e1.record(); torch.cuda.synchronize()
    print('tile', t, '%.1f us' % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)
