"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/pba.h declares,
and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

from photobundle_amd import _lib, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "pba.h")).read()
    declared = sorted(set(re.findall(r"\b(pba_[a-z0-9_]+)\s*\(", header)) - {"pba_allreduce_fn"})
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), "libpba_hip.so does not export %s" % name
    assert sorted(_lib.SYMBOLS) == declared


def test_struct_layouts_match_header_sizes():
    # sizes the C compiler gives the ABI structs (computed from the header field lists)
    import ctypes as C
    assert C.sizeof(_lib.Config) == 4 * 4 + 5 * 8 + 4 * 4
    assert C.sizeof(_lib.StepInfo) == 7 * 8 + 2 * 4
    assert C.sizeof(_lib.IterationSummary) == 4 * 4 + 9 * 8 + 4 * 4 + 5 * 8
    assert C.sizeof(_lib.SolverOptions) == 2 * 4 + 9 * 8 + 2 * 4


def test_default_solver_options_are_the_reference_settings():
    o = engine.default_solver_options()
    # reference src/photobundle.cc:751,756-758 + Ceres defaults
    assert o.max_num_iterations == 500
    assert o.function_tolerance == o.gradient_tolerance == o.parameter_tolerance == 1e-6
    assert o.initial_trust_region_radius == 1e4 and o.min_relative_decrease == 1e-3
    assert o.min_lm_diagonal == 1e-6 and o.max_lm_diagonal == 1e32 and o.jacobi_scaling == 1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.EngineError, match="no HIP device"):
        engine.Engine(64, 64, (100.0, 100.0, 32.0, 32.0), 2, 4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "photobundle_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".cc")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text, \
                    "%s mentions the oracle: the product path must not touch test infrastructure" % f
