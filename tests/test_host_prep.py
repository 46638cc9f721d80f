"""Host-side C of the product (htk_amd/host/prep.c) against the oracle: gConst, inverse variances, log
weights, minimum model durations; accuracy of the device LAdd table."""
import ctypes as C

import numpy as np


def test_prep_functions_bit_equal(native, oracle):
    L, O = native.lib(), oracle.lib()
    rng = np.random.default_rng(3)
    var = rng.uniform(1e-3, 30, size=(50, 39)).astype(np.float32)
    var[0, 0] = 1e-35; var[1, 1] = 1e35                      # clamps of ConvDiagC
    a = np.empty_like(var); b = np.empty_like(var)
    L.htkamd_host_conv_diagc(C.c_size_t(var.size), var.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p))
    O.orc_conv_diagc(C.c_int(var.size), var.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
    assert np.array_equal(a, b)
    for g in range(50):
        x = np.zeros(1, np.float32); y = np.zeros(1, np.float32)
        L.htkamd_host_fix_diag_gconst(C.c_int(39), var[g].ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
        O.orc_fix_diag_gconst(C.c_int(39), var[g].ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
        assert x[0] == y[0]
    L.htkamd_host_mix_log_weight.restype = C.c_float
    L.htkamd_host_mix_log_weight.argtypes = [C.c_float]
    for w in (0.0, 1e-6, 0.99e-5, 1e-5, 0.3, 1.0):
        assert L.htkamd_host_mix_log_weight(w) == O.orc_mix_log_weight(w)


def test_min_dur(native, oracle):
    from htk_amd import synth
    pk, *_ = synth.make_topo_set()
    L, O = native.lib(), oracle.lib()
    want = [3, 0, 1, 1]                                     # 5-state chain, tee model, 4-state with a 2->4 skip, 3-state
    for t in range(int(pk["numTrans"])):
        tp = np.ascontiguousarray(pk["transP"][pk["transOff"][t]:pk["transOff"][t + 1]], np.float32)
        N = int(pk["transN"][t])
        a = L.htkamd_host_min_dur(C.c_int(N), tp.ctypes.data_as(C.c_void_p))
        b = O.orc_min_dur(C.c_int(N), tp.ctypes.data_as(C.c_void_p))
        assert a == b == want[t]


def test_ladd_table_accuracy(native):
    """x + table(d) must agree with x + log(1+exp(d)) to the rounding of the libm pair (2e-16 absolute)."""
    L = native.lib()
    n = L.htkamd_host_ladd_table_size()
    tab = np.empty(n, np.float64)
    L.htkamd_host_build_ladd_table(tab.ctypes.data_as(C.c_void_p))
    deg = 10; inv_h = 4
    assert n == 93 * (deg + 1)
    rng = np.random.default_rng(1)
    d = -rng.uniform(0, 23.0258, 200000)
    k = (-d * inv_h).astype(int)
    r = d + (k + 0.5) / inv_h
    rows = tab.reshape(-1, deg + 1)[k]
    f = rows[:, deg].copy()
    for j in range(deg - 1, -1, -1):
        f = f * r + rows[:, j]
    ref = np.log1p(np.exp(d.astype(np.longdouble))).astype(np.float64)
    assert np.abs(f - ref).max() < 4e-16
