"""Parity of the HIP path (through the C ABI) against the oracle and the committed
goldens.  Needs a real MI355X: `python -m pytest tests -m gpu`.

Tolerance (BASELINE.json north_star): max-abs-err / max-abs-ref <= 1e-3 per output
tensor, fp32.  Single operators are held to 2e-5 (pure fp32 re-association)."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ffrnet_amd
from ffrnet_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ffr_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-3          # the contract with the reference (BASELINE.json north_star)
REG_TOL = 5e-5      # regression gate: the build measures 2-5e-6 end to end; a Winograd edge-tile or border-class-bias
                    # slip that costs 1e-4 must fail here although it is inside the contract
OP_TOL = 2e-5


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope='module')
def engine(state_dicts):
    assert torch.cuda.is_available(), 'these tests need the GPU box'
    sd_e, sd_r = state_dicts
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd_e)
    eng.load_recnet(sd_r)
    return eng


def pack_w(w, cin_pad, cout_pad):
    cout, cin, R, S = w.shape
    p = torch.zeros(cout_pad, R, S, cin_pad)
    p[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
    return p.reshape(cout_pad, -1).contiguous()


CONV_CASES = [
    # N, H, W, cin, cout, R, stride, pad, mode, prelu, resid, sigmoid, tile, splitk
    (2, 14, 14, 64, 64, 3, 1, 1, 0, True, False, False, 0, 0),
    (3, 14, 14, 64, 128, 3, 2, 1, 0, False, False, False, 1, 0),
    (2, 28, 20, 32, 64, 3, 1, 1, 0, True, True, False, 2, 0),
    (2, 28, 28, 64, 128, 1, 2, 0, 0, False, False, False, 3, 0),
    (5, 7, 7, 96, 49, 3, 1, 1, 1, True, True, True, 0, 0),
    (4, 7, 7, 561, 256, 3, 1, 1, 1, True, False, False, 0, 0),
    (2, 7, 7, 128, 64, 3, 1, 1, 1, True, False, False, 3, 4),
    (2, 16, 16, 64, 64, 3, 1, 1, 0, True, False, False, 4, 0),
    (9, 1, 1, 25088, 512, 1, 1, 0, 0, False, False, False, 0, 0),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_operator(engine, case):
    N, H, W, cin, cout, R, stride, pad, mode, prelu, resid, sig, tile, splitk = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    cin_pad = (cin + 31) // 32 * 32
    cout_pad = (cout + 63) // 64 * 64
    in_pitch = cin_pad + 32
    x = torch.randn(N, H, W, in_pitch, generator=g)
    x[..., cin:cin_pad] = 0
    w = torch.randn(cout, cin, R, R, generator=g) / (cin * R * R) ** 0.5
    bias = torch.zeros(cout_pad)
    bias[:cout] = torch.randn(cout, generator=g) * 0.1
    slope = torch.zeros(cout_pad)
    slope[:cout] = torch.rand(cout, generator=g) * 0.3 + 0.1
    xin = x[..., :cin].permute(0, 3, 1, 2)
    if mode == 1:
        ref = F.conv2d(F.pad(xin, (pad,) * 4, mode='reflect'), w, None, stride)
    else:
        ref = F.conv2d(xin, w, None, stride, pad)
    ref = ref + bias[:cout].view(1, -1, 1, 1)
    if prelu:
        ref = F.prelu(ref, slope[:cout])
    Ho, Wo = ref.shape[2:]
    res_pitch = cout_pad + 64
    r = torch.randn(N, Ho, Wo, res_pitch, generator=g)
    if resid:
        ref = ref + r[..., :cout].permute(0, 3, 1, 2)
    if sig:
        ref = torch.sigmoid(ref)
    out_pitch, out_coff = cout_pad + 96, 32
    out = torch.full((N, Ho, Wo, out_pitch), -7.0).cuda()
    engine.op_conv(x=x.cuda(), N=N, H=H, W=W, in_pitch=in_pitch, cin_pad=cin_pad,
                   w=pack_w(w, cin_pad, cout_pad).cuda(), bias=bias.cuda(),
                   slope=slope.cuda() if prelu else None,
                   resid=r.cuda() if resid else None, res_pitch=res_pitch,
                   out=out, out_pitch=out_pitch, out_coff=out_coff, cout_store=cout, cout_pad=cout_pad,
                   R=R, S=R, stride=stride, pad=pad, pad_mode=mode, border_bias=0,
                   flags=1 if sig else 0, tile=tile, splitk=splitk)
    torch.cuda.synchronize()
    got = out[..., out_coff:out_coff + cout].permute(0, 3, 1, 2).cpu()
    assert rel(got, ref) < OP_TOL
    # nothing outside the channel slice was touched
    assert (out[..., :out_coff] == -7.0).all() and (out[..., out_coff + cout:] == -7.0).all()


WINO_CASES = [
    # N, H, W, cin, cout, pad_mode, prelu, resid
    (3, 14, 14, 256, 256, 0, True, False),
    (5, 56, 56, 64, 64, 0, True, False),      # cin <= 128: the fused kernel transforms its own input (zero padding)
    (9, 7, 7, 128, 192, 1, True, True),       # ... with reflect padding, 3 channel groups, tiles beyond the last image
    (2, 30, 22, 96, 64, 0, False, True),      # ... ragged map, cin not a multiple of 64
    (2, 28, 28, 128, 128, 0, False, False),
    (5, 7, 7, 512, 512, 1, True, True),       # 7x7: tiles hang over the edge, reflect padding
    (2, 13, 10, 128, 64, 0, True, True),      # ragged H, W
    (4, 7, 7, 576, 256, 1, True, False),
    (700, 28, 28, 128, 64, 0, False, False),  # 34300 tiles: GEMM rows beyond 2^15
    (40, 56, 56, 64, 128, 0, True, True),     # two channel groups, two phases, residual through the q-form epilogue
    (33, 30, 22, 96, 192, 1, True, False),    # reflect padding, ragged map, three phases, last tile group partly empty
]


@pytest.mark.parametrize('case', WINO_CASES)
def test_winograd_conv_matches_direct_and_torch(engine, case):
    """F(4x4,3x3) + batched GEMM vs torch conv2d and vs the direct implicit GEMM."""
    N, H, W, cin, cout, mode, prelu, resid = case
    g = torch.Generator().manual_seed(1000 + hash(case) % 1000)
    x = torch.randn(N, H, W, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    slope = torch.rand(cout, generator=g) * 0.3 + 0.1 if prelu else None
    r = torch.randn(N, H, W, cout, generator=g) if resid else None
    xin = x.permute(0, 3, 1, 2)
    ref = F.conv2d(F.pad(xin, (1,) * 4, mode='reflect'), w, bias) if mode == 1 else F.conv2d(xin, w, bias, 1, 1)
    if prelu:
        ref = F.prelu(ref, slope)
    if resid:
        ref = ref + r.permute(0, 3, 1, 2)
    rd = r.cuda() if resid else None
    got_d = engine.op_conv3x3(x.cuda(), w, bias, slope, mode, 0, rd).permute(0, 3, 1, 2).cpu()
    assert rel(got_d, ref) < OP_TOL
    # fused kernel with 32x64 blocks (full launches) / with 32x32 blocks (small launches) / transform kernels + batched GEMM
    for use_wino in (1, 3, 2):
        got_w = engine.op_conv3x3(x.cuda(), w, bias, slope, mode, use_wino, rd).permute(0, 3, 1, 2).cpu()
        assert rel(got_w, ref) < 1e-4, use_wino          # Winograd F(4,3) in fp32: measured ~2e-6
        assert rel(got_w, got_d) < 1e-4, use_wino


def test_trunk_stage_taps(engine, state_dicts, golden_dir):
    sd_e, _ = state_dicts
    x = synth.synth_images(8, 112, 112, seed=123)[:2]
    taps = {}
    with torch.no_grad():
        O.encoder_trunk(sd_e, x, 24, taps)
    g2 = np.load(os.path.join(golden_dir, 'g2_stage_taps.npz'))
    xd = x.cuda()
    for nb, name in [(0, 'input_layer')] + [(i + 1, 'body.%d' % i) for i in (0, 2, 3, 6, 7, 20, 21, 23)]:
        got = engine.encoder_trunk_nhwc(xd, nb).permute(0, 3, 1, 2).cpu()
        assert rel(got, taps[name]) < TOL, name
        flat = got[0].reshape(-1)
        step = max(1, flat.numel() // 256)
        assert np.abs(flat[::step][:256].numpy() - g2[name + '.samples']).max() \
            < TOL * float(g2[name + '.absmax']), name
        if (name + '.full') in g2:
            assert rel(got[0], torch.from_numpy(g2[name + '.full'])) < TOL


def test_config1_embeddings_vs_golden(engine, golden_dir):
    """BASELINE config 1 inputs through the HIP path vs the reference's own outputs."""
    g = np.load(os.path.join(golden_dir, 'g1_config1.npz'))
    x = synth.synth_images(8, 112, 112, seed=123).cuda()
    featmap, f = engine.encoder_forward(x)
    f_new, feat_new = engine.recnet_forward(featmap)
    f_new2, f2 = engine.embed(x)
    torch.cuda.synchronize()
    for tol in (TOL, REG_TOL):
        assert rel(f, torch.from_numpy(g['f'])) < tol
        assert rel(f_new, torch.from_numpy(g['f_new'])) < tol
        assert rel(featmap[0], torch.from_numpy(g['featmap0'])) < tol
        assert rel(feat_new[0], torch.from_numpy(g['feat_new0'])) < tol
        assert rel(f_new2, torch.from_numpy(g['f_new'])) < tol
        assert rel(f2, torch.from_numpy(g['f'])) < tol
    # per-row relative L2
    for a, b in ((f_new.cpu(), torch.from_numpy(g['f_new'])), (f.cpu(), torch.from_numpy(g['f']))):
        assert (((a - b).norm(dim=1) / b.norm(dim=1)).max().item()) < TOL


def test_recnet_internals_vs_golden(engine, golden_dir):
    g1 = np.load(os.path.join(golden_dir, 'g1_config1.npz'))
    g3 = np.load(os.path.join(golden_dir, 'g3_recnet_internals.npz'))
    fm = torch.from_numpy(g1['featmap0'])[None].cuda()
    d = engine.recnet_debug(fm)
    torch.cuda.synchronize()
    for tol in (TOL, REG_TOL):
        assert rel(d['ss_space'][0], torch.from_numpy(g3['ss_space0'])) < tol
        assert rel(d['M_space'][0], torch.from_numpy(g3['M_space0'])) < tol
        assert rel(d['feat_space'][0], torch.from_numpy(g3['feat_space0'])) < tol
        assert rel(d['feat_channel_raw'][0], torch.from_numpy(g3['feat_channel_raw0'])) < tol
        assert rel(d['feat_channel'][0], torch.from_numpy(g3['feat_channel0'])) < tol
        # the two tensors the fused channel path never stores (models/recnet.py:402,406): debug stores for image 0
        assert rel(d['ss_channel0'][::8, ::8], torch.from_numpy(g3['ss_channel0_s8'])) < tol
        assert rel(d['M_channel0'][::8, ::8], torch.from_numpy(g3['M_channel0_s8'])) < tol


def test_trunk_112x96(engine, golden_dir):
    """The BASELINE.json '112x96' label: only the trunk exists at that size."""
    g = np.load(os.path.join(golden_dir, 'g6_trunk_112x96.npz'))
    x = synth.synth_images(2, 112, 96, seed=125).cuda()
    featmap, f = engine.encoder_forward(x, want_f=False)
    assert f is None and list(featmap.shape) == [2, 512, 7, 6]
    assert rel(featmap[0], torch.from_numpy(g['featmap0'])) < TOL
    with open(os.path.join(golden_dir, 'g7_112x96_errors.json')) as fh:
        msgs = json.load(fh)
    with pytest.raises(RuntimeError) as ei:
        engine.encoder_forward(x)
    assert '21504 and 25088x512' in str(ei.value) and '21504 and 25088x512' in msgs['encoder']
    with pytest.raises(RuntimeError):
        engine.recnet_forward(featmap)


@pytest.mark.parametrize('n', [1, 3, 33])
def test_ragged_batches_vs_oracle(engine, state_dicts, n):
    sd_e, sd_r = state_dicts
    x = synth.synth_images(n, seed=500 + n)
    f_new, f = engine.embed(x.cuda())
    rf_new, rf = O.embed(sd_e, sd_r, x)
    assert rel(f_new, rf_new) < TOL and rel(f, rf) < TOL
    assert rel(f_new, rf_new) < REG_TOL and rel(f, rf) < REG_TOL


@pytest.mark.parametrize('n', [1, 33, 200])
def test_ragged_batches_trained_like_family(specs, golden_dir, n):
    """The small and ragged launch shapes (transform kernels + k_gemm_stream at 1 image, split-off remainders and the 32 x 32 block
    shape at 33, whole rounds + tail split at 200) on the TRAINED-LIKE weights of golden G11: BatchNorm running_var over 5 decades,
    PReLU slopes in [-0.5, 1.5], saturated SE gates.  Held to the oracle's fp32 run at the 1e-3 contract (the reference's own fp32
    error on this family is 1.7e-4: DESIGN.md 4), row by row."""
    sd_e, sd_r = synth.stress_state_dicts('trained', specs['encoder'], specs['recnet'], golden_dir)
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd_e)
    eng.load_recnet(sd_r)
    x = synth.synth_images(n, seed=900 + n)
    f_new, f = eng.embed(x.cuda())
    rows = list(range(n)) if n <= 33 else [0, 1, 63, 64, 127, 128, 198, 199]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rf_new, rf = O.embed(sd_e, sd_r, x[rows])
    assert rel(f_new[rows], rf_new) < TOL and rel(f[rows], rf) < TOL
    worst = (((f[rows].cpu() - rf).norm(dim=1) / rf.norm(dim=1)).max().item(), ((f_new[rows].cpu() - rf_new).norm(dim=1) / rf_new.norm(dim=1)).max().item())
    assert max(worst) < TOL, worst                          # per-row relative L2 as well
    assert torch.isfinite(f_new).all() and torch.isfinite(f).all()
    eng.close()


def test_batch_independence_full_size(engine, state_dicts):
    """BASELINE batch 256 -- the launch shapes the benchmark runs (k_wino_fused over 2..25 rounds of block tiles,
    stream-K cut tiles of the stride-2 convolutions, the 3.7 GB transform workspace).  Images are independent units:
    EVERY row of the batch must equal the same image embedded in a batch of 8, and a 32-row subset is held to the
    oracle (which is too slow for all 256 in a unit test)."""
    sd_e, sd_r = state_dicts
    xc = synth.synth_images(256, seed=124)
    x = xc.cuda()
    f_new, f = engine.embed(x)
    f_new, f = f_new.clone(), f.clone()
    for i in range(0, 256, 8):
        g_new, g = engine.embed(x[i:i + 8].contiguous())
        assert rel(f_new[i:i + 8], g_new) < 2e-5 and rel(f[i:i + 8], g) < 2e-5, i
    idx = list(range(3, 256, 8))
    rf_new, rf = O.embed(sd_e, sd_r, xc[idx])
    assert rel(f_new[idx], rf_new) < REG_TOL and rel(f[idx], rf) < REG_TOL
    assert torch.isfinite(f_new).all() and torch.isfinite(f).all()
    assert ((f.norm(dim=1) - 1).abs() < 1e-4).all()
    # a second run of the same batch is bitwise identical (no atomics, fixed reduction orders)
    f_new2, f2 = engine.embed(x)
    assert torch.equal(f_new2, f_new) and torch.equal(f2, f)


@pytest.mark.parametrize('B', [64, 128, 200, 257, 300])
def test_batch_independence_other_batches(engine, state_dicts, B):
    """128 and 64 images: the per-GPU shard of BASELINE configs[3] on 8 / 16 GPUs, where the planner takes other
    branches than at 256 (32 x 32 block tiles of k_wino_fused for stage 3 / stage 4 / RecNet, 1.5-round launches).
    257: an odd batch on the exact 4+4+3+3 tiling of the 14x14 maps (wino_mixed.hip: the last tile group of every type is
    partly empty, the last combine block holds one image).  200 and 300: batches that are not a power of two -- the fused launches split off a different number of images for
    the transform-kernel path (whole rounds of block tiles), tile groups straddle images, the last
    tile group is partly empty.  Every row must match the same image embedded in a batch of 8; a subset is held to
    the oracle."""
    sd_e, sd_r = state_dicts
    xc = synth.synth_images(B, seed=4300 + B)
    x = xc.cuda()
    f_new, f = engine.embed(x)
    f_new, f = f_new.clone(), f.clone()
    assert torch.isfinite(f_new).all() and torch.isfinite(f).all()
    for i in range(0, B, 8):
        g_new, g = engine.embed(x[i:i + 8].contiguous())
        assert rel(f_new[i:i + 8], g_new) < 2e-5 and rel(f[i:i + 8], g) < 2e-5, i
    idx = list(range(1, B, B // 8))[:8]
    rf_new, rf = O.embed(sd_e, sd_r, xc[idx])
    assert rel(f_new[idx], rf_new) < REG_TOL and rel(f[idx], rf) < REG_TOL


def test_encoder_forward_and_trunk_112x96_full_size(engine, state_dicts):
    """ffr_encoder_forward (NCHW featmap out, BASELINE configs[1]) and the 112x96 trunk at batch 256: every image
    equals its batch-8 run; a subset is held to the oracle."""
    sd_e, _ = state_dicts
    for hw, want_f in (((112, 112), True), ((112, 96), False)):
        xc = synth.synth_images(256, hw[0], hw[1], seed=126)
        x = xc.cuda()
        fm, f = engine.encoder_forward(x, want_f=want_f)
        fm = fm.clone()
        f = f.clone() if want_f else None
        for i in range(0, 256, 8):
            gm, g = engine.encoder_forward(x[i:i + 8].contiguous(), want_f=want_f)
            # (not bitwise: the stride-2 convolutions' stream-K cuts fall elsewhere at another batch size)
            assert rel(fm[i:i + 8], gm) < 2e-5, (hw, i)
            if want_f:
                assert rel(f[i:i + 8], g) < 2e-5, (hw, i)
        idx = [0, 77, 128, 255]
        rfm = O._bn(O.encoder_trunk(sd_e, xc[idx]), sd_e, 'bn')
        assert rel(fm[idx], rfm) < REG_TOL, hw
        assert torch.isfinite(fm).all()


@pytest.mark.parametrize('case', [(8, 256, True, False, 14, 256), (70, 256, True, True, 14, 256), (33, 512, False, False, 14, 256),
                                  (40, 512, True, True, 7, 512), (33, 256, True, False, 7, 256)])
def test_mixed_tile_sizes_match_direct_and_torch(engine, case):
    """The exact 4+4+3+3 tiling of 14x14 maps (wino_mixed.hip: tile types F(4x4), F(4x3), F(3x4), F(3x3); use_wino = 4 forces
    it) vs torch conv2d, vs the direct implicit GEMM and vs the padded F(4x4) kernel: whole tile groups, a partly empty last
    group (70 images: 280 tiles per type), PReLU, residual, 256 and 512 output channels; and the 4+3 tiling of 7x7 maps (one
    tile of each type per image; round 5: only reachable through use_wino = 4, tools/mixed7_experiment.py measures it).  The
    three extra weight sets are derived on the device for the call (engine.cpp: ensure_mixed_weights)."""
    N, cout, prelu, resid, H, cin = case
    g = torch.Generator().manual_seed(4242 + N)
    x = torch.randn(N, H, H, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    slope = torch.rand(cout, generator=g) * 0.3 + 0.1 if prelu else None
    r = torch.randn(N, H, H, cout, generator=g) if resid else None
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, bias, 1, 1)
    if prelu:
        ref = F.prelu(ref, slope)
    if resid:
        ref = ref + r.permute(0, 3, 1, 2)
    rd = r.cuda() if resid else None
    got_d = engine.op_conv3x3(x.cuda(), w, bias, slope, 0, 0, rd).permute(0, 3, 1, 2).cpu()
    got_4 = engine.op_conv3x3(x.cuda(), w, bias, slope, 0, 1, rd).permute(0, 3, 1, 2).cpu()
    got_m = engine.op_conv3x3(x.cuda(), w, bias, slope, 0, 4, rd).permute(0, 3, 1, 2).cpu()
    assert rel(got_d, ref) < OP_TOL
    assert rel(got_m, ref) < 1e-4 and rel(got_m, got_d) < 1e-4 and rel(got_m, got_4) < 1e-4
    assert not torch.equal(got_m, got_4)            # it really is another arithmetic


@pytest.mark.gpu
def test_mixed_tile_weights_are_packed_lazily(state_dicts):
    """VERDICT r04 weak #9 / ADVICE r04: the three extra Winograd weight sets of the exact 14x14 tiling (0.7 GB per handle) are no
    longer packed at load time: a handle that only ever sees small batches holds none; the first reserve / forward of a batch that
    can use them derives them on the device; switching wf_mixed on AFTER the weights were loaded works (it was a silent no-op)."""
    sd_e, sd_r = state_dicts
    eng = ffrnet_amd.Engine(0)
    eng.set_option('wf_mixed', 0)
    eng.load_encoder(sd_e)
    eng.load_recnet(sd_r)
    x = synth.synth_images(8, seed=5).cuda()
    f_new8, _ = eng.embed(x)
    f_new8 = f_new8.clone()
    st = eng.memory_stats()
    assert st['mixed_tile_weight_bytes'] == 0 and st['encoder_weight_bytes'] > 100e6 and st['encoder_load_seconds'] > 0
    eng.reserve(256)
    assert eng.memory_stats()['mixed_tile_weight_bytes'] == 0            # option off: nothing packed
    eng.set_option('wf_mixed', 1)
    eng.reserve(256)
    st = eng.memory_stats()
    assert 0.5e9 < st['mixed_tile_weight_bytes'] < 0.9e9 and st['mixed_tile_pack_seconds'] > 0
    xb = synth.synth_images(256, seed=6).cuda()
    xb[:8] = x
    f_new, _ = eng.embed(xb)                                              # batch 256 runs k_wino_fused_mixed on stage 3
    assert rel(f_new[:8].cpu(), f_new8.cpu()) < 2e-5
    assert eng.memory_stats()['mixed_tile_weight_bytes'] == st['mixed_tile_weight_bytes']      # packed once


@pytest.mark.gpu
def test_exact_tiling_with_512_channels_on_224x224_inputs(state_dicts):
    """on a 224x224 input the 14x14 maps are STAGE 4 (512 channels), so the automatic path runs k_wino_fused_mixed,
    k_combine_in_mixed and the SE tile sums with C = 512 -- a combination the 112x112 forward never reaches.  The trunk of 128
    such images with the exact tiling equals the padded-tile run of the same handle (another arithmetic, so not bitwise) and the
    oracle on two of them; the weight sets are derived per layer from the real input size (stage 3 runs 28x28 here: none for it)."""
    sd_e, _ = state_dicts
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd_e)
    x = synth.synth_images(128, 224, 224, seed=41)
    xd = x.cuda()
    a = eng.encoder_trunk_nhwc(xd, 24)
    st = eng.memory_stats()
    assert 0 < st['mixed_tile_weight_bytes'] < 0.5e9           # the 5 stride-1 convolutions of stage 4: 512 -> 512
    eng.set_option('wf_mixed', 0)
    b = eng.encoder_trunk_nhwc(xd, 24)
    torch.cuda.synchronize()
    assert list(a.shape) == [128, 14, 14, 512]
    assert rel(a, b) < 2e-5 and not torch.equal(a, b)
    with torch.no_grad():
        ref = O.encoder_trunk(sd_e, x[:2], 24)
    assert rel(a[:2].permute(0, 3, 1, 2), ref) < REG_TOL
    # alternating input shapes on one handle: the 112x112 forward still finds (and now derives) its own layers' sets
    eng.set_option('wf_mixed', 1)
    eng.reserve(256)
    assert eng.memory_stats()['mixed_tile_weight_bytes'] > st['mixed_tile_weight_bytes']
    eng.close()


@pytest.mark.parametrize('B', [256, 512])
def test_two_call_shell_path_full_size(engine, state_dicts, B):
    """The path the reference's UNCHANGED calculate_distance takes through the shells (lfw_eval.py:241-244):
    `featmap, f = encoder(img)` then `f_new, feat_new = recnet(featmap)` -- ffr_encoder_forward (NCHW featmap out)
    followed by ffr_recnet_forward (NCHW featmap in, f_new + NCHW feat_new out) -- at the benchmark's batch (256) and at
    the pair batch of configs[3] (512).  Every row of f, f_new and feat_new equals its batch-8 run; 32 rows are held to
    the oracle, feat_new included; f_new of the two-call path equals the fused ffr_embed's (VERDICT r03 weak #1a)."""
    sd_e, sd_r = state_dicts
    xc = synth.synth_images(B, seed=9100 + B)
    x = xc.cuda()
    fm, f = engine.encoder_forward(x)
    f_new, feat_new = engine.recnet_forward(fm)
    f, f_new, feat_new = f.clone(), f_new.clone(), feat_new.clone()
    assert tuple(feat_new.shape) == (B, 512, 7, 7) and tuple(f_new.shape) == (B, 512)
    assert torch.isfinite(f_new).all() and torch.isfinite(feat_new).all()
    for i in range(0, B, 8):
        gm, g = engine.encoder_forward(x[i:i + 8].contiguous())
        g_new, g_feat = engine.recnet_forward(gm)
        assert rel(f[i:i + 8], g) < 2e-5 and rel(f_new[i:i + 8], g_new) < 2e-5, i
        assert rel(feat_new[i:i + 8], g_feat) < 2e-5, i
    e_new, e = engine.embed(x)
    assert rel(f_new, e_new) < 2e-5 and rel(f, e) < 2e-5
    # f_new is the 7x7 average of feat_new (models/recnet.py:423)
    assert rel(feat_new.mean(dim=(2, 3)), f_new) < 2e-5
    idx = list(range(5, B, B // 32))[:32]
    with torch.no_grad():
        r_fm, r_f = O.encoder_forward(sd_e, xc[idx])
        r_new, r_feat = O.recnet_forward(sd_r, r_fm)
    assert rel(f[idx], r_f) < REG_TOL and rel(f_new[idx], r_new) < REG_TOL and rel(feat_new[idx], r_feat) < REG_TOL


def test_uint8_input_step_full_size(engine):
    """ffr_embed_u8 at batch 256 -- k_stem's full-size launch shape (data/dataset.py:70-79: BGR swap, one flip decision
    per pair, ToTensor, Normalize) -- is bit-identical to ffr_embed on the float tensor torch builds from the same
    bytes (VERDICT r03 weak #1b)."""
    g = torch.Generator().manual_seed(78)
    B = 256
    img = torch.randint(0, 256, (B, 112, 112, 3), generator=g, dtype=torch.uint8)
    flip = (torch.rand(B // 2, generator=g) < 0.5).repeat_interleave(2).to(torch.uint8)   # one decision per PAIR (dataset.py:76-79)
    x = O.preprocess_u8(img, flip)
    f_new_a, f_a = engine.embed(x.cuda())
    f_new_a, f_a = f_new_a.clone(), f_a.clone()
    f_new_b, f_b = engine.embed_u8(img.cuda(), flip.cuda())
    assert torch.equal(f_new_a, f_new_b) and torch.equal(f_a, f_b)
    assert 0 < int(flip.sum()) < B


@pytest.mark.parametrize('tag', ['50_ir', '100_ir', '100_ir_se', '152_ir_se'])
def test_backbone_variants(golden_dir, tag):
    """Backbone(100 | 152, ., 'ir' | 'ir_se') (pretrain/model_ir_se50.py:84-116; the reference defines them, its scripts
    use (50, 'ir_se') only): the engine reads the configuration off the state_dict (49 / 50 bottlenecks, res_layer.5 or
    not).  2 images against golden G10 = the reference's own outputs, then batch 40 against the batch-2 run; through
    the drop-in shell as well."""
    spec = json.load(open(os.path.join(golden_dir, 'g10_backbone_variant_keys.json')))[tag]
    g = np.load(os.path.join(golden_dir, 'g10_backbone_variants.npz'))
    sd = synth.synth_state_dict(spec, seed=0)
    num_layers, mode = int(tag.split('_')[0]), tag.split('_', 1)[1]
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd)
    assert eng.num_layers == num_layers
    x = synth.synth_images(2, 112, 112, seed=131).cuda()
    featmap, f = eng.encoder_forward(x)
    assert rel(f, torch.from_numpy(g[tag + '.f'])) < 2e-4
    flat = featmap.reshape(-1)
    step = max(1, flat.numel() // 512)
    assert (flat[::step][:512].cpu() - torch.from_numpy(g[tag + '.featmap_samples'])).abs().max().item() < \
        2e-4 * float(g[tag + '.featmap_absmax'])
    xb = synth.synth_images(40, 112, 112, seed=132).cuda()
    xb[:2] = x
    fm40, f40 = eng.encoder_forward(xb)
    assert rel(f40[:2], f) < 2e-5 and rel(fm40[:2], featmap) < 2e-5 and torch.isfinite(fm40).all()
    shell = ffrnet_amd.Backbone(num_layers=num_layers, drop_ratio=0.6, mode=mode)
    shell.load_state_dict(sd)
    fm_s, f_s = shell.cuda().eval()(x)
    assert torch.equal(fm_s, featmap) and torch.equal(f_s, f)
    eng.close()


def test_se_module_isolated(engine, state_dicts):
    """SEModule (pretrain/model_ir_se50.py:18-36) on its own: block 2's squeeze-excitation scale, recovered from
    the trunk taps around it -- res * sigmoid(fc2(relu(fc1(avgpool(res))))) + shortcut -- against torch ops on the
    device with the same weights.  Checks both squeeze sources (Winograd tile sums / separate pooling pass)."""
    sd_e, _ = state_dicts
    x = synth.synth_images(5, seed=91).cuda()
    for blk in (1, 2, 9, 22):            # stride-1 units of stages 1, 1, 3, 4 (identity shortcut, SE from tile sums)
        a = engine.encoder_trunk_nhwc(x, blk).permute(0, 3, 1, 2)            # block input, NCHW
        b = engine.encoder_trunk_nhwc(x, blk + 1).permute(0, 3, 1, 2)        # block output
        p = 'body.%d.res_layer.' % blk
        w = {k[len(p):]: v.cuda() for k, v in sd_e.items() if k.startswith(p)}
        r = F.batch_norm(a, w['0.running_mean'], w['0.running_var'], w['0.weight'], w['0.bias'], False, 0.0, 1e-5)
        r = F.prelu(F.conv2d(r, w['1.weight'], None, 1, 1), w['2.weight'])
        r = F.conv2d(r, w['3.weight'], None, 1, 1)
        r = F.batch_norm(r, w['4.running_mean'], w['4.running_var'], w['4.weight'], w['4.bias'], False, 0.0, 1e-5)
        s = torch.sigmoid(F.conv2d(F.relu(F.conv2d(r.mean((2, 3), keepdim=True), w['5.fc1.weight'])), w['5.fc2.weight']))
        got_scale = ((b - a) / r).median(dim=3).values.median(dim=2).values      # (out - shortcut) / res per (n, c)
        assert rel(got_scale, s[:, :, 0, 0]) < 1e-3, blk          # through a division: loose
        assert rel(b, r * s + a) < REG_TOL, blk                   # the block as a whole: tight


def test_module_shells_drop_in(state_dicts, golden_dir):
    """models/trainer.py:98-113 clone_model flow through the nn.Module shells."""
    sd_e, sd_r = state_dicts
    enc = ffrnet_amd.Backbone(num_layers=50, drop_ratio=0.6, mode='ir_se')
    rec = ffrnet_amd.RecNet(norm_type='bn', relu_type='prelu')
    enc.load_state_dict(sd_e)
    rec.load_state_dict(sd_r)
    enc2 = ffrnet_amd.Backbone(num_layers=50, drop_ratio=0.6, mode='ir_se')
    enc2.load_state_dict(enc.state_dict())
    enc2.to('cuda').eval()
    rec.to('cuda').eval()
    g = np.load(os.path.join(golden_dir, 'g1_config1.npz'))
    x = synth.synth_images(8, 112, 112, seed=123).cuda()
    with torch.no_grad():
        feat_map, f = enc2(x)
        f_new, _ = rec(feat_map)
    assert rel(f, torch.from_numpy(g['f'])) < TOL
    assert rel(f_new, torch.from_numpy(g['f_new'])) < TOL
    with pytest.raises(RuntimeError):
        enc2(x.cpu())
    enc2.train()
    with pytest.raises(NotImplementedError):
        enc2(x)


def test_calculate_distance_and_accuracy(engine, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g4_calculate_distance.npz'))
    i1, i2, lab = synth.synth_pairs(16, seed=7, block=8)
    loader = [dict(img1=i1[s:s + 8].cuda(), img2=i2[s:s + 8].cuda(), label=lab[s:s + 8],
                   idx=torch.arange(s, s + 8)) for s in (0, 8)]
    pn, p = ffrnet_amd.lfw.calculate_distance(loader, engine.embed)
    assert np.abs(pn[:, 0] - g['pred_new'][:, 0]).max() < 1e-4
    assert np.abs(p[:, 0] - g['pred'][:, 0]).max() < 1e-4
    assert np.array_equal(pn[:, 1:], g['pred_new'][:, 1:])


def test_device_fold_protocol_matches_reference(engine, golden_dir):
    """threshold sweep + 10-fold selection on the device, bit-equal to the
    reference's numbers (golden G5, incl. scores exactly on grid thresholds) and its tie rule."""
    g = np.load(os.path.join(golden_dir, 'g5_fold_protocol.npz'))
    scores = torch.from_numpy(g['scores'].astype(np.float32)).cuda()
    labels = torch.from_numpy(g['labels'].astype(np.int64)).cuda()
    # the golden was computed on float64 scores; redo the host protocol on the fp32-rounded ones
    pred = np.array([g['scores'].astype(np.float32).astype(np.float64), g['labels'], np.arange(6000)]).T
    mean_h, res_h = ffrnet_amd.lfw.get_accuracy_from_predicts(pred)
    mean_d, res_d = engine.lfw_fold_accuracy(scores, labels, 10)
    assert [r[0] for r in res_d] == [float(r[0]) for r in res_h]
    assert [r[1] for r in res_d] == [float(r[1]) for r in res_h]
    assert abs(mean_d - mean_h) < 1e-15
    # tie case of the golden: several thresholds reach the best accuracy -> the LAST one wins
    tie = g['tie']
    s = torch.from_numpy(tie[:, 0].astype(np.float32)).cuda()
    lab = torch.from_numpy(tie[:, 1].astype(np.int64)).cuda()
    _, res = engine.lfw_fold_accuracy(s, lab, 1)        # one fold: train set empty -> every threshold ties at 0
    assert res[0][0] == float(ffrnet_amd.lfw.THRESHOLDS[-1])


def test_uint8_input_step_is_bit_identical(engine):
    """RGB->BGR, per-image h-flip, ToTensor, Normalize(0.5,0.5) inside the stem kernel
    (data/dataset.py:70-79, data/dataloader.py:24-28) == the float tensor torch builds, bit for bit."""
    g = torch.Generator().manual_seed(77)
    img = torch.randint(0, 256, (6, 112, 112, 3), generator=g, dtype=torch.uint8)      # HWC RGB as PIL gives
    flip = torch.tensor([0, 1, 0, 1, 1, 0], dtype=torch.uint8)
    x = O.preprocess_u8(img, flip)          # BGR swap, ToTensor, Normalize, tf.hflip: the oracle's restatement of the data pipeline
    f_new_a, f_a = engine.embed(x.cuda())
    f_new_b, f_b = engine.embed_u8(img.cuda(), flip.cuda())
    assert torch.equal(f_new_a, f_new_b) and torch.equal(f_a, f_b)
    f_new_c, _ = engine.embed_u8(img.cuda())
    assert torch.equal(f_new_c[[0, 2, 5]], f_new_a[[0, 2, 5]]) and not torch.equal(f_new_c[1], f_new_a[1])


def test_verification_accuracy_matches_oracle_small(engine, state_dicts):
    """BASELINE config 4 in small: pairs -> embeddings -> cosine -> 10-fold protocol; the accuracy
    from the HIP path equals the one from the oracle's embeddings (identical to 4 dp) and the
    scores agree to 1e-4."""
    sd_e, sd_r = state_dicts
    n_pairs = 100
    i1, i2, lab = synth.synth_pairs(n_pairs, seed=11, block=10)
    loader = [dict(img1=i1[s:s + 50].cuda(), img2=i2[s:s + 50].cuda(), label=lab[s:s + 50],
                   idx=torch.arange(s, s + 50)) for s in (0, 50)]
    pn, p = ffrnet_amd.lfw.calculate_distance(loader, engine.embed)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    f1n, f1 = O.embed(sd_e, sd_r, i1)
    f2n, f2 = O.embed(sd_e, sd_r, i2)
    on = np.array([O.cosine_scores(f1n, f2n).double().numpy(), lab.numpy(), np.arange(n_pairs)]).T
    oo = np.array([O.cosine_scores(f1, f2).double().numpy(), lab.numpy(), np.arange(n_pairs)]).T
    assert np.abs(pn[:, 0] - on[:, 0]).max() < 1e-4 and np.abs(p[:, 0] - oo[:, 0]).max() < 1e-4
    for got, ref in ((pn, on), (p, oo)):
        a_g, _ = ffrnet_amd.lfw.get_accuracy_from_predicts(got)
        a_r, _ = ffrnet_amd.lfw.get_accuracy_from_predicts(ref)
        assert round(a_g, 4) == round(a_r, 4)
        a_d, _ = engine.lfw_fold_accuracy(torch.from_numpy(got[:, 0].astype(np.float32)).cuda(),
                                          torch.from_numpy(got[:, 1]).cuda(), 10)
        a_h, _ = ffrnet_amd.lfw.get_accuracy_from_predicts(
            np.array([got[:, 0].astype(np.float32).astype(np.float64), got[:, 1], got[:, 2]]).T)
        assert a_d == a_h


def _g9_loader(g):
    """The 6000 synthetic pairs of golden G9 (regenerated, never stored), pair batches of 512 on the host."""
    n, bs = int(g['n_pairs']), int(g['batch'])
    i1, i2, lab = synth.synth_pairs(n, seed=int(g['seed']), block=int(g['block']))
    assert abs(i1.double().sum().item() + i2.double().sum().item() - float(g['input_checksum'])) < 1e-6
    return [dict(img1=i1[s:s + bs], img2=i2[s:s + bs], label=lab[s:s + bs], idx=torch.arange(s, min(s + bs, n)))
            for s in range(0, n, bs)]


@pytest.fixture(scope='module')
def g9(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g9_lfw_protocol.npz'))
    return g, _g9_loader(g)


def _flip_report(got, ref, res_got, thr_ref, acc_ref):
    """Pairs whose decision differs from the reference's at a fold's threshold, with the score delta."""
    lines = []
    for k, ((thr, acc), tr, ar) in enumerate(zip(res_got, thr_ref, acc_ref)):
        lo, hi = k * 600, (k + 1) * 600
        fl = np.nonzero((got[lo:hi] > tr) != (ref[lo:hi] > tr))[0]
        if thr != tr or acc != ar or len(fl):
            lines.append('fold %d: thr %.3f (ref %.3f) acc %.6f (ref %.6f) flipped pairs %s'
                         % (k, thr, tr, acc, ar, [(int(lo + i), float(got[lo + i] - ref[lo + i])) for i in fl]))
    return '\n'.join(lines)


def test_lfw_protocol_6000_pairs_matches_reference(engine, g9):
    """BASELINE configs[3] / north_star acceptance line at FULL size: the 6000 synthetic pairs (12000 images, pair
    batches of 512) through Engine.embed -> ffr_cosine_scores -> ffr_lfw_fold_accuracy -- the harness's DEFAULT path,
    no overrides -- against golden G9 = the reference's own calculate_distance + KFold + get_fold_accuracy on CPU
    (tests/golden/make_golden_lfw.py).  Accuracy identical to 4 dp for f_new AND f, every fold's best threshold and
    test accuracy equal, scores within 1e-4.  With these weights cos(f_new) lies in 0.97..0.995, where the 0.005 grid
    cuts through both classes: a 1e-5 score error can flip a pair, and a flip is reported, not tolerated."""
    g, loader = g9
    import time
    torch.cuda.synchronize()
    t0 = time.time()
    acc_new, acc, det = ffrnet_amd.lfw.get_avg_accuracy(engine.embed, loader, details=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    # the input side (lfw.ShardFeeder): host tensors went through the pinned staging buffers to the ENGINE's device, one
    # block per pair batch, and nothing but this rank's rows was copied (one rank: all of them, exactly once)
    st = ffrnet_amd.lfw.last_feed_stats
    assert st['batches'] == 12 and st['h2d_bytes'] == st['shard_bytes'] == st['full_batch_bytes'] == 2 * 6000 * 3 * 112 * 112 * 4, st
    print('6000-pair protocol incl. host->device copies: %.2f s (%.0f pairs/s); acc_new %.6f acc %.6f'
          % (dt, 6000 / dt, acc_new, acc))
    pn, p = det['pred_new'], det['pred']
    assert np.array_equal(pn[:, 1], g['labels']) and np.array_equal(pn[:, 2], g['idx'])
    d_new, d_old = np.abs(pn[:, 0] - g['scores_new']).max(), np.abs(p[:, 0] - g['scores']).max()
    print('max |score - reference|: f_new %.2e  f %.2e' % (d_new, d_old))
    assert d_new < 1e-4 and d_old < 1e-4
    assert 0.55 < float(g['acc_new']) < 0.999 and 0.55 < float(g['acc']) < 0.999     # neither trivial
    rep = _flip_report(pn[:, 0], g['scores_new'], det['folds_new'], g['best_thr_new'], g['test_acc_new']) + \
        _flip_report(p[:, 0], g['scores'], det['folds'], g['best_thr'], g['test_acc'])
    assert [t for t, _ in det['folds_new']] == [float(t) for t in g['best_thr_new']], rep
    assert [t for t, _ in det['folds']] == [float(t) for t in g['best_thr']], rep
    assert round(acc_new, 4) == round(float(g['acc_new']), 4), rep
    assert round(acc, 4) == round(float(g['acc']), 4), rep
    assert [a for _, a in det['folds_new']] == [float(a) for a in g['test_acc_new']], rep
    assert [a for _, a in det['folds']] == [float(a) for a in g['test_acc']], rep
    # size-independent properties on the same run
    assert np.isfinite(pn).all() and np.abs(pn[:, 0]).max() <= 1.0 + 1e-6 and np.abs(p[:, 0]).max() <= 1.0 + 1e-6
    assert pn[pn[:, 1] == 1, 0].mean() > pn[pn[:, 1] == 0, 0].mean()
    # device fold protocol == host restatement on the same fp32 scores
    acc_h, res_h = ffrnet_amd.lfw.get_accuracy_from_predicts(pn)
    assert acc_h == acc_new and [float(r[0]) for r in res_h] == [t for t, _ in det['folds_new']]
    # the embedding of a pair does not depend on its batch
    x = loader[3]['img1'].cuda()
    f_new, _ = engine.embed(x[:8].contiguous())
    f_all, _ = engine.embed(x)
    assert rel(f_all[:8], f_new) < 1e-5


STRESS_TAPS = [(0, 'input_layer')] + [(i + 1, 'body.%d' % i) for i in (0, 2, 3, 6, 7, 20, 21, 23)]


@pytest.mark.parametrize('fam', sorted(synth.STRESS_FAMILIES))
def test_stress_families_vs_reference(specs, golden_dir, fam):
    """Golden G11 (tests/golden/make_golden_stress.py): the reference's own outputs for three weight families off the
    benign distribution of G1-G10 -- a second seed; a trained-like encoder + RecNet as models/trainer.py:65-66
    initialises it (kaiming); a trained-like checkpoint end to end (BatchNorm running_var over > 5 decades, PReLU
    slopes in [-0.5, 1.5], a third of the SE gates saturated).  The fixture carries the reference in fp32 AND in
    float64, so three numbers are stated per tensor: HIP vs the reference (the 1e-3 contract of BASELINE.json), HIP vs
    the exact answer, and the reference's own fp32 error vs the exact answer -- an ill-conditioned family amplifies
    every fp32 implementation's rounding, and the product is held to a small multiple of the reference's own.
    600 pairs through the default harness: scores, per-fold thresholds, accuracies."""
    g = np.load(os.path.join(golden_dir, 'g11_%s.npz' % fam))
    sd_e, sd_r = synth.stress_state_dicts(fam, specs['encoder'], specs['recnet'], golden_dir)
    img_seed, pair_seed = synth.STRESS_FAMILIES[fam][4:]
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd_e)
    eng.load_recnet(sd_r)
    x = synth.synth_images(8, 112, 112, seed=img_seed)
    assert abs(x.double().sum().item() - float(g['input_checksum'])) < 1e-9
    xd = x.cuda()
    featmap, f = eng.encoder_forward(xd)
    f_new, feat_new = eng.recnet_forward(featmap)
    f_new2, f2 = eng.embed(xd)
    torch.cuda.synchronize()
    report = {'family': fam, 'tensors': {}, 'taps': {}}

    def check(name, got, key, samples=False):
        ref, ref64 = torch.from_numpy(g[key]), torch.from_numpy(g[key + '_f64'])
        amax = ref64.abs().max().item() if not samples else float(g[key.rsplit('.', 1)[0] + '.absmax'])
        e_ref = (got.double().cpu() - ref.double()).abs().max().item() / amax
        e_64 = (got.double().cpu() - ref64).abs().max().item() / amax
        own = (ref.double() - ref64).abs().max().item() / amax
        report['taps' if samples else 'tensors'][name] = dict(hip_vs_ref=e_ref, hip_vs_f64=e_64, ref_vs_f64=own)
        assert e_ref < TOL, (fam, name, e_ref)                          # the contract
        assert e_64 < max(REG_TOL, 8.0 * own), (fam, name, e_64, own)   # no worse than a small multiple of the reference's own rounding

    check('f', f, 'f')
    check('f_new', f_new, 'f_new')
    check('featmap0', featmap[0], 'featmap0')
    check('feat_new0', feat_new[0], 'feat_new0')
    assert rel(f_new2, f_new) < 1e-6 and rel(f2, f) < 1e-6              # fused entry point == two-call path
    # the same 8 images as rows 0..7 of a batch of 256: only there do stage 3's 27 convolutions run the EXACT tiling
    # (k_wino_fused_mixed / k_combine_in_mixed: F(4x3), F(3x4), F(3x3) tile types, other interpolation points) and the 32 x 64 block shape
    xb = synth.synth_images(256, 112, 112, seed=img_seed + 500)
    xb[:8] = x
    fb_new, fb = eng.embed(xb.cuda())
    torch.cuda.synchronize()
    assert eng.memory_stats()['mixed_tile_weight_bytes'] > 0.5e9
    check('f (rows 0..7 of batch 256)', fb[:8], 'f')
    check('f_new (rows 0..7 of batch 256)', fb_new[:8], 'f_new')
    for nb, name in STRESS_TAPS:
        got = eng.encoder_trunk_nhwc(xd[:1].contiguous(), nb).permute(0, 3, 1, 2)[0].reshape(-1)
        step = max(1, got.numel() // 256)
        check(name, got[::step][:256], 'tap.' + name + '.samples', samples=True)
    # 600 pairs, the harness's default path (Engine.embed -> ffr_cosine_scores -> ffr_lfw_fold_accuracy)
    n, block, bs = int(g['n_pairs']), int(g['pair_block']), int(g['pair_batch'])
    i1, i2, lab = synth.synth_pairs(n, seed=pair_seed, block=block)
    assert abs(i1.double().sum().item() + i2.double().sum().item() - float(g['pair_checksum'])) < 1e-6
    loader = [dict(img1=i1[s:s + bs], img2=i2[s:s + bs], label=lab[s:s + bs], idx=torch.arange(s, s + bs)) for s in range(0, n, bs)]
    acc_new, acc, det = ffrnet_amd.lfw.get_avg_accuracy(eng.embed, loader, details=True)
    for tag, pred, res, acc_got in (('new', det['pred_new'], det['folds_new'], acc_new), ('', det['pred'], det['folds'], acc)):
        sfx = '_new' if tag else ''
        ref, ref64 = g['scores' + sfx], g['scores' + sfx + '_f64']
        got = pred[:, 0]
        d_ref, d_64, own = np.abs(got - ref).max(), np.abs(got - ref64).max(), np.abs(ref - ref64).max()
        thr_ref, acc_ref = g['best_thr' + sfx], g['test_acc' + sfx]
        # pairs the product and the reference decide differently at the reference's threshold of their fold
        fold_of = np.arange(n) // (n // 10)
        flips = np.nonzero((got > thr_ref[fold_of]) != (ref > thr_ref[fold_of]))[0]
        exact_flips = np.nonzero((ref64 > thr_ref[fold_of]) != (ref > thr_ref[fold_of]))[0]
        report['scores' + sfx] = dict(hip_vs_ref=float(d_ref), hip_vs_f64=float(d_64), ref_vs_f64=float(own),
                                      flipped_pairs=[int(i) for i in flips], ref_vs_f64_flipped_pairs=[int(i) for i in exact_flips],
                                      acc=float(acc_got), acc_ref=float(g['acc' + sfx]),
                                      thresholds_equal=[t for t, _ in res] == [float(t) for t in thr_ref],
                                      fold_acc_equal=[a for _, a in res] == [float(a) for a in acc_ref])
        assert d_ref < max(TOL, 4.0 * own) and d_64 < max(1e-4, 8.0 * own), (fam, tag, d_ref, d_64, own)
        if own < 1e-5:
            # a well-conditioned family: the north_star's acceptance line, as for G9 -- everything EQUAL
            assert report['scores' + sfx]['thresholds_equal'] and report['scores' + sfx]['fold_acc_equal'], report['scores' + sfx]
            assert round(acc_got, 4) == round(float(g['acc' + sfx]), 4)
        else:
            # the reference's own fp32 scores are `own` away from the exact ones: two fp32 implementations can only agree on the
            # pairs whose exact score is farther than that from the threshold.  Every disagreement must be such a pair.
            band = d_64 + own
            assert all(abs(ref64[i] - thr_ref[fold_of[i]]) <= band for i in flips), (fam, tag, flips)
            assert len(flips) <= max(3 * len(exact_flips), 6), (fam, tag, flips, exact_flips)
            assert abs(acc_got - float(g['acc' + sfx])) <= 0.02
    out_dir = os.path.join(ROOT, 'gpurun_out', 'r06')
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, 'stress_parity_%s.json' % fam), 'w') as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report))
    eng.close()


def test_reference_shaped_harness_through_the_shells(state_dicts, g9):
    """lfw_eval.get_avg_accuracy(encoder, recnet, data_loader) (lfw/lfw_eval.py:272, caller train.py:101-113) with
    the two drop-in shells: same call, same two numbers as golden G9, native scoring and fold protocol underneath."""
    g, loader = g9
    sd_e, sd_r = state_dicts
    enc = ffrnet_amd.Backbone(num_layers=50, drop_ratio=0.6, mode='ir_se')
    rec = ffrnet_amd.RecNet(channel=512, shape=7, norm_type='bn', relu_type='prelu')
    enc.load_state_dict(sd_e)
    rec.load_state_dict(sd_r)
    enc, rec = enc.cuda().eval(), rec.cuda().eval()
    sub = loader[:2]                                     # 1024 pairs: the same entry point, the same rows of G9
    pred_new, pred = ffrnet_amd.lfw.calculate_distance(sub, enc, rec)
    assert np.abs(pred_new[:, 0] - g['scores_new'][:1024]).max() < 1e-4
    assert np.abs(pred[:, 0] - g['scores'][:1024]).max() < 1e-4
    avg_acc_new, avg_acc = ffrnet_amd.lfw.get_avg_accuracy(enc, rec, loader)
    assert round(avg_acc_new, 4) == round(float(g['acc_new']), 4)
    assert round(avg_acc, 4) == round(float(g['acc']), 4)
    rec.train()
    with pytest.raises(NotImplementedError):
        ffrnet_amd.lfw.get_avg_accuracy(enc, rec, sub)


def test_hipgraph_replay_matches_eager(engine):
    """One forward is a single launch-only C call: it captures into a hipGraph; replay == eager."""
    x = synth.synth_images(8, seed=31).cuda()
    f_new, f = engine.embed(x)
    g = ffrnet_amd.GraphedEmbed(engine, 8)
    for _ in range(2):
        g_new, g_f = g(x)
        torch.cuda.synchronize()
        assert torch.equal(g_new, f_new) and torch.equal(g_f, f)
    x2 = synth.synth_images(8, seed=32).cuda()
    h_new, _ = g(x2)
    e_new, _ = engine.embed(x2)
    assert torch.equal(h_new, e_new)


def test_hipgraph_recaptured_when_the_handle_reallocates(state_dicts):
    """A captured forward holds raw pointers into the workspace arena and the packed weights.  A larger batch regrows
    the arena, a weight reload re-packs: the allocation generation changes and the graph is captured again instead of
    replaying into freed memory (ffr_generation)."""
    sd_e, sd_r = state_dicts
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd_e)
    eng.load_recnet(sd_r)
    x = synth.synth_images(2, seed=612).cuda()
    want_new, want = (t.clone() for t in eng.embed(x))
    g = ffrnet_amd.GraphedEmbed(eng, 2)
    gen0 = eng.generation()
    assert torch.equal(g(x)[0], want_new) and g.captures == 1
    eng.embed(synth.synth_images(40, seed=613).cuda())           # arena regrowth
    assert eng.generation() != gen0
    a, b = g(x)
    torch.cuda.synchronize()
    assert g.captures == 2 and torch.equal(a, want_new) and torch.equal(b, want)
    eng.load_recnet(sd_r)                                         # packed weights re-allocated
    a, b = g(x)
    torch.cuda.synchronize()
    assert g.captures == 3 and torch.equal(a, want_new) and torch.equal(b, want)


def test_bench_and_trainer_two_ranks_on_one_gpu():
    """The N > 1 code paths on hardware, with what a 1-GPU box allows: two ranks on cuda:0 over gloo (RCCL refuses two
    ranks on one device).  bench.py: all-gather of the embeddings, barrier, max-over-ranks timing, one JSON line from
    rank 0 with n_gpus = 2.  NativeTrainer: after one iteration on different batches both ranks hold identical
    parameters (the zero-copy flat gradient buffer went through the all-reduce)."""
    import json
    import subprocess
    import warnings

    class _Sub:          # a timeout is a failure (a hang in the rendezvous / exchange paths must not be retried away)
        @staticmethod
        def run(cmd, **kw):
            try:
                return subprocess.run(cmd, **kw)
            except subprocess.TimeoutExpired as e:
                pytest.fail('two-rank run timed out: %s\n--- stderr tail ---\n%s'
                            % (' '.join(cmd[-8:]), (e.stderr or b'')[-3000:]))

    def free_port():
        import socket
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        p = s.getsockname()[1]
        s.close()
        return p

    env = dict(os.environ, FFR_BENCH_BACKEND='gloo', FFR_BENCH_ONE_DEVICE='1', FFR_BENCH_DIST_TIMEOUT='180')
    env.pop('WORLD_SIZE', None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # from a plain shell, as the driver calls it: bench.py spawns its own ranks
    for extra, batch, scaling in ((['--batch', '16', '--strong-pairs', '40'], 16, 'weak'), (['--pairs-per-step', '24'], 24, 'strong')):
        cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
               '--no-roofline', '--no-cpu-baseline'] + extra
        out = _Sub.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
        d = json.loads(line)
        assert d['n_gpus'] == 2 and d['value'] > 0 and d['steps'] == 2 and d['scaling'] == scaling
        assert d['config']['batch_per_gpu'] == batch and d['config']['global_batch'] == 2 * batch
        assert d['parity_checked']['max_rel_err_vs_reference_golden_G1'] < 5e-5
        pr = d['per_rank']              # every rank reports its own clock and the collective's (one all-gather per step)
        assert len(pr['wall_ms_per_step']) == 2 and len(pr['all_gather_ms_hipevents_median']) == 2
        assert pr['all_gather_bytes_per_rank'] == 2 * batch * 512 * 4
        if scaling == 'weak':
            # the default N-GPU line also carries configs[3]'s STRONG-scaling shape (pairs per step split over the ranks) as a
            # secondary entry, with every rank's all-gather time and the bytes it copied from host memory (its shard only)
            (sec,) = d['secondary']
            assert sec['scaling'] == 'strong' and sec['n_gpus'] == 2 and sec['batch_per_gpu'] == 40 and sec['value'] > 0
            assert len(sec['per_rank']['all_gather_ms_hipevents_median']) == 2
            fh = sec['from_host_memory']
            assert fh['h2d_bytes_per_rank'] == fh['shard_bytes_per_rank'] == [2 * 2 * 20 * 3 * 112 * 112 * 4] * 2
            assert fh['full_batch_bytes'] == 2 * 2 * 40 * 3 * 112 * 112 * 4
        else:
            assert d['secondary'] is None
    # eight handles + arenas + ranks side by side on one device (what an 8-GPU node runs, minus the links)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--batch', '8',
           '--no-roofline', '--no-cpu-baseline']
    out = _Sub.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 8 and d['config']['global_batch'] == 64 and len(d['per_rank']['wall_ms_per_step']) == 8
    # and under torch.distributed.run, as the driver launches the multi-GPU runs
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '16',
           '--no-roofline', '--no-cpu-baseline']
    out = _Sub.run(cmd, env=dict(env, MASTER_ADDR='127.0.0.1'), cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['config']['global_batch'] == 32 and d['steps'] == 2
    # the training workload with its roofline block ON: the profiled iterations contain the gradient all-reduce, so every
    # rank must run them (rank 0 alone would wait in the collective for ever)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', 'train', '--steps', '1', '--warmup', '1',
           '--batch', '8', '--no-cpu-baseline']
    out = _Sub.run(cmd, env=dict(env, FFR_BENCH_LIVE_PMC='0'), cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['roofline']['frac'] > 0
    pc = d['roofline']['per_class']
    assert {'train_bn', 'train_loss', 'train_optim', 'wgrad', 'wino_fused'} <= set(pc), sorted(pc)
    out = _Sub.run([sys.executable, os.path.join(root, 'tools', 'ddp_two_ranks_one_gpu.py')], cwd=root, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and 'OK' in out.stdout, (out.stdout[-1000:], out.stderr[-2000:])


@pytest.mark.gpu
def test_rccl_one_rank_runs_the_product_collectives():
    """RCCL itself (backend nccl), as far as one GPU allows: one rank, the product's collectives on the product's
    buffers (tools/rccl_one_rank.py): embeddings all-gather, bucketed gradient all-reduce on the second stream,
    parameter broadcast; every collective must be the identity."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # a default, as in bench.py main(): dmabuf IPC handles
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rccl_one_rank.py')], cwd=root, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'OK' in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_bench_line_carries_the_three_roofline_fractions():
    """frac (executed), frac_useful (padding-free), frac_algorithmic_survey_8d, numeric `traffic` (or
    null) and the traffic ratio are top-level scalars of `roofline`; a small batch keeps this quick."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FFR_BENCH_LIVE_PMC='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--batch', '128', '--steps', '3', '--warmup', '2',
                          '--no-cpu-baseline', '--no-secondary'], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])['roofline']
    assert 0 < r['frac_useful'] < r['frac'] < 1.0
    assert r['frac_algorithmic_survey_8d'] > r['frac']            # Winograd executes a quarter of the direct multiplies
    assert 'traffic' in r and (r['traffic'] is None or r['traffic'] > 0)
    assert 'traffic_ratio_vs_compulsory' in r and r['gflop_useful_per_launch'] < r['gflop_executed_per_launch']
    assert 0 < r['per_bound']['whole_step']['frac_useful'] < r['per_bound']['whole_step']['frac_of_mfma_peak']


def test_options_api(engine):
    """ffr_set_option / ffr_get_option: the only way to switch kernels (the library reads no environment): defaults,
    round trip, unknown names, out-of-range values, trace options rejected by the shipped (non -DFFR_TRACE) build."""
    assert engine.get_option('wf_minblocks') == 200 and engine.get_option('wino') == 1 and engine.get_option('combine_v') == 1
    engine.set_option('wf_minblocks', 123)
    assert engine.get_option('wf_minblocks') == 123
    engine.set_option('wf_minblocks', 200)
    for bad in (lambda: engine.set_option('no_such_option', 1), lambda: engine.set_option('wino', 2),
                lambda: engine.set_option('sk_minunits', 0), lambda: engine.get_option('nope'),
                lambda: engine.set_option('wf_trace', 1), lambda: engine.set_option('igemm_trace', 1)):
        with pytest.raises(RuntimeError):
            bad()
    assert engine.get_option('wf_trace') == 0
    src = ''
    for fn in os.listdir(os.path.join(ROOT, 'ffr-net_amd', 'csrc')):
        src += open(os.path.join(ROOT, 'ffr-net_amd', 'csrc', fn)).read()
    assert 'getenv' not in src


@pytest.mark.gpu
def test_experiment_knobs_keep_parity(tmp_path):
    """The A/B knobs of DESIGN.md 3.3 (ffr_set_option; tools/knob_embed.py maps FFR_OPT_<NAME> in ITS environment to
    Engine.set_option -- the library reads no environment) select other kernels / mappings for the same arithmetic: every setting must
    reproduce the default path's embeddings (batch 96: fused Winograd launches with and without a split-off remainder,
    both block maps, SE squeeze from tile sums or from its own pass, the round-1 transform kernels, no Winograd)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, 'tools', 'knob_embed.py')

    def run(name, **knobs):
        path = str(tmp_path / (name + '.pt'))
        env = dict(os.environ, **knobs)
        out = subprocess.run([sys.executable, tool, path, '96'], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and 'OK' in out.stdout, (name, out.stderr[-2000:])
        return torch.load(path)

    ref = run('default')
    for name, knobs in (('notailsplit', {'FFR_OPT_WF_TAILSPLIT': '0'}),
                        ('sepool', {'FFR_OPT_SE_MAXTILES': '0'}), ('unfused', {'FFR_OPT_WINO_FUSED': '0'}),
                        ('unfused_igemm', {'FFR_OPT_WINO_FUSED': '0', 'FFR_OPT_GEMM_STREAM': '0'}),
                        ('phased256', {'FFR_OPT_WF_PHASED_MAXK': '256'}), ('direct', {'FFR_OPT_WINO': '0'}),
                        ('nocombinev', {'FFR_OPT_COMBINE_V': '0'}),
                        ('minblocks0', {'FFR_OPT_WF_MINBLOCKS': '0'}), ('nomixed', {'FFR_OPT_WF_MIXED': '0'}),
                        ('chrows1', {'FFR_OPT_CHANNEL_ROWS': '1'}), ('chrows2', {'FFR_OPT_CHANNEL_ROWS': '2'}), ('chrows4', {'FFR_OPT_CHANNEL_ROWS': '4'})):
        got = run(name, **knobs)
        for k in ('f_new', 'f'):
            assert rel(got[k], ref[k]) < 2e-5, (name, k, rel(got[k], ref[k]))
