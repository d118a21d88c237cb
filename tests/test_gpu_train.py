"""GPU parity of the RecNet training step through the C ABI (include/ffrnet_train.h):
single operators against torch CPU autograd, the whole step against the oracle (oracle/ffr_oracle_train.py)
and the golden G8 captured from the reference's own Trainer code."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ffrnet_amd
from ffrnet_amd import synth

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))

GRAD_TOL = 2e-4     # gradients: max-abs-err / max-abs-ref per tensor (fp32 re-association over <= 12544-row sums)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope='module')
def engine():
    return ffrnet_amd.Engine(0)


CONVLAYER_CASES = [   # (G, N, cin, cout)
    (1, 4, 64, 64), (2, 3, 128, 128), (1, 5, 49, 49), (2, 2, 561, 256), (1, 3, 128, 49), (1, 2, 1024, 512),
]


@pytest.mark.parametrize('case', CONVLAYER_CASES)
def test_convlayer_train_forward_backward(engine, case):
    """reflect-pad -> conv3x3 -> BatchNorm2d(train) -> PReLU, models/recnet.py:78-85: outputs, batch / running
    statistics, data gradient and all four parameter gradients vs torch autograd on the CPU."""
    G, N, cin, cout = case
    g = torch.Generator().manual_seed(77 + cin + cout)
    x = torch.randn(G * N, 7, 7, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    gamma = torch.rand(cout, generator=g) * 0.5 + 0.75
    beta = torch.randn(cout, generator=g) * 0.1
    slope = torch.rand(cout, generator=g) * 0.3 + 0.1
    da = torch.randn(G * N, 7, 7, cout, generator=g)
    res = engine.op_convlayer_train(x.cuda(), G, w, gamma, beta, slope, da.cuda())
    torch.cuda.synchronize()
    # torch reference, one BatchNorm batch per group
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr, gr, br, sr = (t.clone().requires_grad_(True) for t in (w, gamma, beta, slope))
    rm, rv = torch.zeros(cout), torch.zeros(cout)
    outs = []
    for gi in range(G):
        y = F.conv2d(F.pad(xr[gi * N:(gi + 1) * N], (1,) * 4, mode='reflect'), wr)
        outs.append(F.prelu(F.batch_norm(y, rm, rv, gr, br, True, 0.1, 1e-5), sr))
    out = torch.cat(outs)
    out.backward(da.permute(0, 3, 1, 2))
    assert rel(res['out'].permute(0, 3, 1, 2), out) < 2e-5
    assert rel(res['running_mean'], rm) < 2e-5 and rel(res['running_var'], rv) < 2e-5
    assert rel(res['dx'].permute(0, 3, 1, 2), xr.grad) < GRAD_TOL
    assert rel(res['dw'], wr.grad) < GRAD_TOL
    assert rel(res['dgamma'], gr.grad) < GRAD_TOL
    assert rel(res['dbeta'], br.grad) < GRAD_TOL
    assert rel(res['dslope'], sr.grad) < GRAD_TOL
