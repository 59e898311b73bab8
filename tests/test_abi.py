"""The C-ABI library loads on a machine without a GPU and exports every symbol include/dia_hip.h
declares; argument validation paths that never launch a kernel behave as documented."""
import ctypes
import os
import re

import pytest

from dia_hip import binding as hb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "dia_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dia_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    L = hb.lib()
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dia_hip.h but not exported"
    assert set(names) == set(hb.EXPORTS)
    assert L.dia_abi_version() == hb.ABI_VERSION


def test_struct_sizes_match_header():
    # the C side is compiled from the same header; a trivial way to pin the mirror is sizeof via a
    # tiny C program, compiled with the system compiler (no GPU involved)
    import subprocess, tempfile
    prog = r'''
    #include <stdio.h>
    #include "dia_hip.h"
    int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(dia_gemm_args), sizeof(dia_attn_args),
        sizeof(dia_embed_args), sizeof(dia_sample_args), sizeof(dia_dec_layer), sizeof(dia_engine_desc), sizeof(dia_enc_attn_args),
        sizeof(dia_dec_prefill_args), sizeof(dia_seg_args)); return 0; }
    '''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    mine = [ctypes.sizeof(t) for t in (hb.GemmArgs, hb.AttnArgs, hb.EmbedArgs, hb.SampleArgs, hb.DecLayer, hb.EngineDesc, hb.EncAttnArgs, hb.DecPrefillArgs, hb.SegArgs)]
    assert sizes == mine


def test_argument_validation_without_gpu():
    L = hb.lib()
    g = hb.GemmArgs()
    assert L.dia_gemm(ctypes.byref(g), None) == -1          # DIA_E_ARG: null pointers
    assert b"null" in L.dia_last_error()
    a = hb.AttnArgs()
    assert L.dia_attn(ctypes.byref(a), None) == -1
    s = hb.SampleArgs()
    assert L.dia_sample(ctypes.byref(s), None) == -1
    sg = hb.SegArgs()
    assert L.dia_seg_mlp(ctypes.byref(sg), None) == -1
    assert L.dia_seg_workspace_bytes() > L.dia_seg_workspace_control_bytes() > 0
    with pytest.raises(hb.DiaHipError):
        hb.check(-1, "x")
