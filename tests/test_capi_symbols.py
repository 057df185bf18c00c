"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/nns.h declares (no compute calls without a GPU); the product never imports the oracle."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, PKG


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'nns.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(nns_[a-z0-9_]+)\s*\(', txt)))


def test_header_declares_what_the_binding_binds():
    from nns import _lib
    assert declared_symbols() == _lib.exported_names()


def test_library_loads_and_exports_every_declared_symbol():
    from nns import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert _lib.lib().nns_version() == 1          # major 0, minor 1
    assert _lib.lib().nns_last_error() is not None


def test_missing_library_fails_loudly(monkeypatch):
    from nns import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libnns_hip.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.lib()


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fs in os.walk(PKG):
        for f in fs:
            if f.endswith(('.py', '.hip', '.cpp', '.h')):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M) or 'oracle/_ref' in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_reference_import_paths_resolve_to_nns():
    import src.boundary as B
    import nns.boundary
    assert B.DirichletBoundaryCondition is nns.boundary.DirichletBoundaryCondition
    bc = B.NeumannBoundaryCondition(0.5, 'left', 0.1, 0.2)
    assert (bc.type, bc.boundary, bc.value, bc.dx, bc.dy) == ('neumann', 'left', 0.5, 0.1, 0.2)
    with pytest.raises(AssertionError):
        B.DirichletBoundaryCondition(0, 'left', 1, 0.2)          # dx must be float (src/boundary.py:17)
    with pytest.raises(AssertionError):
        B.DirichletBoundaryCondition(0, 'front', 0.1, 0.2)


def test_mirrors_have_no_cpu_fallback():
    """The neural mirrors' compute exists as HIP kernels only: CPU tensors must raise, not run an eager stand-in."""
    import pytest
    import torch
    from nns.neural_spectral import spectral_ode, spectral_rnn
    from nns.neural_spectral.anode import odesolver
    m = spectral_ode.PDEFunc(2, 4, 4)
    t = torch.arange(3) + 1
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 4, 4), t)                       # ODEFunc integration off-device
    with pytest.raises(RuntimeError):
        spectral_ode.expand(torch.zeros(3, 2, 3), torch.zeros(2, 3, 16))
    with pytest.raises(RuntimeError):
        spectral_rnn.PDEFunc(2, 4, 4)(torch.zeros(1, 3, 4, 4), t)
    with pytest.raises(RuntimeError):
        odesolver(spectral_ode.ODEFunc(6), torch.zeros(1, 6), {'Nt': 2, 'method': 'RK4'})
    # an arbitrary user callable still goes through the reference's generic stepper (API parity, not a kernel fallback)
    out = odesolver(lambda t_, y: -y, torch.ones(1, 2), {'Nt': 4, 'method': 'RK4'})
    assert out.shape == (4, 1, 2) and abs(out[-1, 0, 0].item() - 0.3679) < 1e-3


def test_call_recorder_records_and_replays_c_calls():
    """nns._lib.CallRecorder (the slab step's replay of recorded C-ABI calls on fixed buffers): calls made inside a stage are recorded with their
    exact arguments and repeated by replay(); a failing status on replay raises like the direct call.  CPU only: the recorded call is one that
    fails its argument check before any HIP call."""
    import ctypes as C
    from nns import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    rec = _lib.CallRecorder()
    ptrs = (C.c_void_p * 3)(None, None, None)
    with rec.stage(('chunk', 0)):
        rc = _lib.lib().nns_slab_pack_halo_f32(ptrs, 3, None, None, None, 4, 3, 2, 8, 64, 4, None)        # g0 + Bc > B, NULL buffers: refused
        assert rc == -1
        assert _lib.lib().nns_version() == 1                                                           # queries are not recorded
    assert not isinstance(_lib.lib(), _lib.CallRecorder._Proxy)                                        # the proxy is gone after the stage
    assert [name for _, _, name in rec.stages[('chunk', 0)]] == ['nns_slab_pack_halo_f32']
    with pytest.raises(_lib.NnsError, match='nns_slab_pack_halo_f32'):
        rec.replay(('chunk', 0))
    with rec.stage(('chunk', 0)):                                                                      # re-recording a stage replaces it
        pass
    rec.replay(('chunk', 0))
