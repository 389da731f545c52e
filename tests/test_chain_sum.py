"""Exactness of the parallel evaluation of sequential fp64 sums (fnn_chain.h) on the CPU
model of the GPU block structure: every result must equal the scalar loop bit for bit,
including inputs built to hit ties, binade crossings, negative / huge / subnormal terms
and (guard band off) mispredicted binades that force the fallback paths."""
import ctypes as C

import numpy as np
import pytest


def _fns(emu_api):
    lib = emu_api.lib
    lib.emu_chain_model.restype = C.c_double
    lib.emu_chain_model.argtypes = [C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.emu_chain_model2.restype = C.c_double
    lib.emu_chain_model2.argtypes = [C.POINTER(C.c_double), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.emu_chain_serial.restype = C.c_double
    lib.emu_chain_serial.argtypes = [C.POINTER(C.c_double), C.c_int32]
    return lib


def chain_cases():
    rng = np.random.default_rng(7)
    cases = {}
    for m in (1, 2, 31, 32, 33, 1000, 4097, 32768, 40000):
        cases[f"uniform_{m}"] = rng.random(m) + 2.0 ** -10
    cases["half_zeros"] = np.where(rng.random(30000) < 0.5, 0.0, rng.random(30000))
    cases["ties_coarse"] = rng.integers(0, 8, 20000) * 2.0 ** -3 + 2.0 ** 40 * (np.arange(20000) == 0)
    cases["ties_half_ulp"] = np.concatenate([[1.0], np.full(5000, 2.0 ** -53), rng.integers(0, 4, 5000) * 2.0 ** -53])
    cases["dec4"] = np.floor(rng.random(32768) * 1e4 + 1) / 1e4
    cases["wide_magnitudes"] = np.exp(rng.uniform(-40, 40, 20000))
    cases["growing"] = 1.5 ** np.arange(1500)
    cases["negatives"] = rng.standard_normal(10000)
    cases["few_negatives"] = np.where(rng.random(20000) < 0.001, -rng.random(20000), rng.random(20000))
    cases["subnormal"] = np.concatenate([rng.integers(1, 1000, 3000) * 5e-324, rng.random(3000) * 1e-300])
    cases["huge_to_inf"] = np.full(3000, 1e306)
    cases["with_nan"] = np.concatenate([rng.random(1000), [np.nan], rng.random(1000)])
    cases["power_of_two_sums"] = np.full(32768, 2.0 ** -5)
    cases["all_zero"] = np.zeros(5000)
    # sequential rounding drifts away from the exact sum: every add of 0.75 ulp rounds up to a
    # whole ulp, so the running sum reaches 2.0 after 64 adds while the exact (and the
    # predicted) prefix is still below 2.0 -> with the guard band off the binade is mispredicted
    u = 2.0 ** -52
    cases["drift_crossing"] = np.concatenate([[2.0 - 64 * u], np.full(400, 0.75 * u), rng.random(500)])
    cases["drift_crossing_long"] = np.concatenate([[2.0 - 4096 * u], np.full(8000, 0.75 * u), rng.random(500)])
    cases["leading_zeros"] = np.concatenate([np.zeros(3000), rng.random(3000)])
    return cases


@pytest.mark.parametrize("ept", [8, 16, 32])
@pytest.mark.parametrize("guard", [22, 0])
def test_chain_model_is_bit_exact(emu_api, ept, guard):
    lib = _fns(emu_api)
    seen = np.zeros(4, dtype=np.int64)
    for name, v in chain_cases().items():
        v = np.ascontiguousarray(v, dtype=np.float64)
        p = v.ctypes.data_as(C.POINTER(C.c_double))
        st = (C.c_int32 * 4)()
        got = lib.emu_chain_model(p, len(v), ept, guard, st)
        ref = lib.emu_chain_serial(p, len(v))
        a, b = np.array([got]).view(np.int64)[0], np.array([ref]).view(np.int64)[0]
        assert a == b or (np.isnan(got) and np.isnan(ref)), (name, got, ref, list(st))
        seen += np.array(list(st))
    assert seen[0] > 0 and seen[1] > 0  # composed runs and one-by-one chunks both occurred
    assert seen[3] < 1000000  # the integer form of the run update agreed with the fp64 form everywhere
    if guard == 0:
        assert seen[2] > 0 and seen[3] > 0  # mispredicted runs / threads were rejected and redone one by one


@pytest.mark.parametrize("guard", [22, 0])
def test_record_form_is_bit_exact(emu_api, guard):
    """The second form (one special addend per chunk + merged constants, fnn_chain.h "records"): same inputs, same bar;
    on the engine-like inputs almost no chunk is left to be added one by one."""
    lib = _fns(emu_api)
    seen = np.zeros(4, dtype=np.int64)
    for name, v in chain_cases().items():
        v = np.ascontiguousarray(v, dtype=np.float64)
        p = v.ctypes.data_as(C.POINTER(C.c_double))
        st = (C.c_int32 * 4)()
        got = lib.emu_chain_model2(p, len(v), guard, st)
        ref = lib.emu_chain_serial(p, len(v))
        a, b = np.array([got]).view(np.int64)[0], np.array([ref]).view(np.int64)[0]
        assert a == b or (np.isnan(got) and np.isnan(ref)), (name, got, ref, list(st))
        seen += np.array(list(st))
        if name == "uniform_32768" and guard:
            assert st[1] <= 3 and st[2] == 0, list(st)  # (the first chunk crosses several binades; no fallback)
    assert seen[0] > 0 and seen[1] > 0
    if guard == 0:
        assert seen[2] > 0  # mispredictions were caught by the verification and went through the first form
