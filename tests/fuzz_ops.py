"""Random operators for the fuzz tests of the table records / grouped diagonal terms (csrc/plan.h: DevTab, DevPass::gbucket)."""
import numpy as np


def shared_mask_operator(rs, L, parity):
    """An operator made to SHARE masks: for a handful of sets of 1-4 flipped spins (2 or 4 when a Parity sector is wanted),
    5-24 Pauli strings each -- X or Y at every flipped spin, a Z string from a small pool elsewhere, so that a mask's terms
    fall into one group or several, of one term or many -- and a random long-range ZZ diagonal on top."""
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, op_sum, op_product
    terms = []
    for _ in range(int(rs.randint(3, 9))):
        S = sorted(rs.choice(L, size=int(rs.choice([2, 4])) if parity else int(rs.randint(1, 5)), replace=False).tolist())
        others = [i for i in range(L) if i not in S]
        pool = [[i for i in others if rs.rand() < 0.4] for _ in range(int(rs.randint(1, 4)))]
        for _ in range(int(rs.randint(5, 25))):
            z = pool[int(rs.randint(len(pool)))]
            ops = [(sigmay if rs.rand() < 0.5 else sigmax)(i) for i in S] + [sigmaz(i) for i in z]
            terms.append(float(rs.uniform(-1, 1)) * op_product(ops))
    for _ in range(int(rs.randint(0, 40))):                      # diagonal terms across the tile's edge
        i, j = rs.choice(L, size=2, replace=False).tolist()
        terms.append(float(rs.uniform(-1, 1)) * sigmaz(i) * sigmaz(j))
    H = op_sum(terms)
    H.L = L
    return H
