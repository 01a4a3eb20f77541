"""Model helpers used by the benchmark Hamiltonians (reference ``src/dynamite/extras.py``)."""
from .operators import sigmax, sigmay, sigmaz, index_product


def commutator(op1, op2):
    """[O1, O2]  (extras.py:4-19)."""
    return op1 * op2 - op2 * op1


def majorana(idx):
    """Majorana fermion as a spin-chain boundary (extras.py:22-59): sigma_z on
    every spin below floor(idx/2), then sigma_x (idx even) or sigma_y (idx odd)."""
    b_idx = idx // 2
    rtn = sigmay(b_idx) if idx % 2 else sigmax(b_idx)
    if b_idx > 0:
        rtn = index_product(sigmaz(), size=b_idx) * rtn
    return rtn
