#!/usr/bin/env python3
"""Round-5 experiment: how many of a phase's 36 patch values should k_wino_fused<1,.> request under the previous phase's last K
chunk (NPRE; 16 since round 3)?  Builds the library with -DFFR_WF_NPRE=N into build/npre<N>/ (only wino_fused.hip differs) and, on a
GPU, times one forward per variant in its own process.
    python tools/npre_experiment.py build 16 24 32        (no GPU needed)
    python tools/npre_experiment.py run 16 24 32          (via gpurun)"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def build(n):
    d = os.path.join(ROOT, 'build', 'npre%d' % n)
    os.makedirs(d, exist_ok=True)
    g.build()
    flags = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++20', '-Wno-unused-value', '-I', os.path.join(ROOT, 'include'),
             '-ffile-prefix-map=%s=.' % ROOT, '-DFFR_WF_NPRE=%d' % n, '-cuid=wino_fused_hip']
    o = os.path.join(d, 'wino_fused.hip.o')
    subprocess.check_call([g.HIPCC] + flags + ['-c', os.path.join(g.CSRC, 'wino_fused.hip'), '-o', o])
    objs = [o if s == 'wino_fused.hip' else os.path.join(ROOT, 'build', s + '.o') for s in g.SOURCES]
    lib = os.path.join(d, 'libffrnet_hip.so')
    subprocess.check_call([g.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    return lib


if sys.argv[1] == 'build':
    for n in sys.argv[2:]:
        print(build(int(n)))
elif sys.argv[1] == 'run':
    for rep in range(2):
        for n in sys.argv[2:]:
            lib = os.path.join(ROOT, 'build', 'npre%s' % n, 'libffrnet_hip.so')
            code = ("import sys; sys.path.insert(0, %r); from ffrnet_amd import native; native.set_library(%r); "
                    "import runpy; sys.argv = ['time_embed']; runpy.run_path(%r, run_name='__main__')"
                    % (ROOT, lib, os.path.join(ROOT, 'tools', 'time_embed.py')))
            out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if 'ms/forward' in l]
            print('NPRE %s: %s' % (n, line[-1] if line else out.stderr[-400:]))
