#!/usr/bin/env python3
"""Build the diagnostics variant of the library (-DFFR_TRACE) and make this process use it.

    import trace_build; trace_build.use()      # before the first ffrnet_amd.Engine

The shipped libffrnet_hip.so contains no trace code and reads no environment; the in-kernel clock stamps of
k_wino_fused / k_igemm exist only in build/trace/libffrnet_hip_trace.so, behind the options "wf_trace" /
"igemm_trace" (Engine.set_option)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def use():
    import __graft_entry__ as g
    path = g.build(trace=True)
    from ffrnet_amd import native
    native.set_library(path)
    return path


if __name__ == '__main__':
    import __graft_entry__ as g
    print(g.build(trace=True))
