"""CPU tests of the drop-in boundary: the shared library loads, exports exactly the symbols
include/vqwave.h declares, the ctypes structs match the C structs, and argument validation
returns errors (no compute without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, 'include', 'vqwave.h')


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vqw_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg._lib
    lib = L.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), 'libvqwave.so does not export %s' % n
    assert sorted(L.SIGNATURES) == names
    assert lib.vqw_abi_version() == 1


def test_ctypes_structs_match_c_layout(pkg, tmp_path):
    L = pkg._lib
    prog = tmp_path / 'sz.c'
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vqwave.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                    'sizeof(vqw_conv_desc), offsetof(vqw_conv_desc, x0), sizeof(vqw_wgrad_desc), offsetof(vqw_wgrad_desc, p),'
                    'sizeof(vqw_ar_weights), offsetof(vqw_ar_weights, post2_w),'
                    'sizeof(vqw_f16x3_gate_desc), sizeof(vqw_f16x3_out_desc), offsetof(vqw_f16x3_out_desc, aux0_kc0),'
                    'sizeof(vqw_f16x3_sconv_desc), sizeof(vqw_f16x3_wgrad_desc), offsetof(vqw_f16x3_wgrad_desc, q_planes),'
                    'offsetof(vqw_f16x3_wgrad_desc, p_tap_chunk));return 0;}\n')
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(prog), '-o', str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(L.ConvDesc), L.ConvDesc.x0.offset, ctypes.sizeof(L.WgradDesc), L.WgradDesc.p.offset,
            ctypes.sizeof(L.ArWeights), L.ArWeights.post2_w.offset,
            ctypes.sizeof(L.F16x3GateDesc), ctypes.sizeof(L.F16x3OutDesc), L.F16x3OutDesc.aux0_kc0.offset,
            ctypes.sizeof(L.F16x3SconvDesc), ctypes.sizeof(L.F16x3WgradDesc), L.F16x3WgradDesc.q_planes.offset,
            L.F16x3WgradDesc.p_tap_chunk.offset]
    assert got == want


def test_errors_are_reported_not_raised_in_c(pkg):
    L = pkg._lib
    lib = L.lib()
    assert lib.vqw_conv_gemm(None, None) != 0
    assert b'null descriptor' in lib.vqw_last_error()
    d = L.ConvDesc()
    d.B, d.T_out, d.T_in, d.M, d.C0, d.ntaps, d.in_stride, d.ldw = 1, 64, 64, 16, 24, 1, 1, 16
    assert lib.vqw_conv_gemm(ctypes.byref(d), None) != 0
    assert b'multiples of 16' in lib.vqw_last_error()
    assert lib.vqw_mu_law_encode_f32(None, None, 4, None) != 0
    assert lib.vqw_rowsum(None, None, None, None, 1.0, 1, 1, 4, 0, None) != 0
    assert lib.vqw_ar_decode_create(None, None, 1) != 0
    with pytest.raises(RuntimeError, match='libvqwave'):
        L.check(1)


def test_python_front_end_refuses_cpu_tensors(pkg):
    import torch
    with pytest.raises(ValueError, match='GPU'):
        pkg.kernels.mu_law_encode_f32(torch.zeros(4))
