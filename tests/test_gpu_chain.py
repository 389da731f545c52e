"""GPU: the block-parallel evaluation of sequential fp64 sums must equal the scalar loop
bit for bit on the same adversarial inputs as the CPU model test."""
import ctypes as C

import numpy as np
import pytest

from test_chain_sum import chain_cases

pytestmark = pytest.mark.gpu


def serial_sum(v):
    s = 0.0
    for x in v.tolist():
        s += x
    return s


@pytest.mark.parametrize("ept", [32])
@pytest.mark.parametrize("guard", [22, 0])
def test_gpu_chain_sum_is_bit_exact(hip_api, ept, guard):
    seen = np.zeros(4, dtype=np.int64)
    for name, v in chain_cases().items():
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = C.c_double(0.0)
        st = (C.c_int32 * 4)()
        hip_api.check(hip_api.test_chain_sum(0, v.ctypes.data_as(C.POINTER(C.c_double)), len(v), guard, ept,
                                             C.byref(out), st))
        ref = serial_sum(v)
        a = np.array([out.value]).view(np.int64)[0]
        b = np.array([ref]).view(np.int64)[0]
        assert a == b or (np.isnan(out.value) and np.isnan(ref)), (name, out.value, ref, list(st))
        seen += np.array(list(st))
    assert seen[0] > 0 and seen[1] > 0
    if guard == 0:
        assert seen[2] > 0 and seen[3] > 0
