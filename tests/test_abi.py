"""CPU: the C-ABI shared library loads without a GPU and exports every symbol include/madqp.h
declares; the product refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

import madqp_jl_amd as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "madqp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(madqp_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = M.load_cdll()
    syms = header_symbols()
    assert len(syms) >= 50
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/madqp.h but not exported"
    assert sorted(M.EXPORTED_SYMBOLS) == syms, "ctypes prototypes out of sync with the header"
    assert lib.madqp_version() == 100


def test_state_struct_layout_matches_header():
    hdr = open(os.path.join(ROOT, "include", "madqp.h")).read()
    body = hdr[hdr.index("typedef struct madqp_state {"):hdr.index("} madqp_state;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\*?\s*([a-z_]+)\s*[,;]", body.split("{", 1)[1])
    from madqp_jl_amd._lib import CState

    assert [f[0] for f in CState._fields_] == names
    assert ctypes.sizeof(CState) == 8 * len(names)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    with pytest.raises(M.MadQPError):
        M.HipBackend(0)


def test_usage_errors_are_return_codes_not_aborts():
    lib = M.load_cdll()
    assert lib.madqp_ctx_create(0, None, None) == -1  # MADQP_ERR_ARG
    assert lib.madqp_ctx_destroy(None) == 0
    assert lib.madqp_gemv(None, 0, 1, 1, 1.0, None, 1, None, 0.0, None) == -1
    assert lib.madqp_chol_solve(None, None) == -1
    assert lib.madqp_last_error(None) == b"null context"


def test_julia_glue_binds_only_exported_symbols_with_the_header_layout():
    """julia/MadQPHIP.jl cannot run here (no Julia): at least every ccall target must exist in the library, and its
    CState must list the fields of madqp_state in the header's order (the GPU replay is tests/test_gpu_julia_replay.py)."""
    src = open(os.path.join(ROOT, "julia", "MadQPHIP.jl")).read()
    code = "\n".join(line.split("#", 1)[0] for line in src.splitlines())
    bound = set(re.findall(r"(?::|@k )(madqp_[a-z0-9_]+)", code))
    assert len(bound) >= 25 and bound <= set(M.EXPORTED_SYMBOLS), sorted(bound - set(M.EXPORTED_SYMBOLS))
    body = code[code.index("struct CState"):]
    body = body[:body.index("\nend")]
    fields = re.findall(r"([a-z_]+)::", body)
    from madqp_jl_amd._lib import CState

    assert fields == [f[0] for f in CState._fields_]
    # the plugin surface SURVEY.md 8b lists: every method the reference calls on the KKT system / linear solver
    for method in ("MadNLP.create_kkt_system", "MadNLP.num_variables", "MadNLP.get_jacobian", "MadNLP.get_hessian",
                   "MadNLP.is_inertia_correct", "MadNLP.initialize!", "MadNLP.compress_jacobian!",
                   "MadNLP.compress_hessian!", "MadNLP.jtprod!", "MadNLP.build_kkt!", "MadNLP.solve!(kkt::HIPKKTSystem",
                   "mul!(w::MadNLP.AbstractKKTVector", "MadNLP.factorize!(s::HIPCholeskySolver",
                   "MadNLP.solve!(s::HIPCholeskySolver", "MadNLP.introduce", "MadNLP.default_options",
                   "MadNLP.is_supported", "MadIPM.is_factorized", "HIPCholeskySolver(aug_com::HIPDenseKKTMatrix"):
        assert method in src, method
