#!/usr/bin/env python
"""
The reference's flagship large-scale example on this engine: ground state and gap of the nearest-neighbour Heisenberg
model on a kagome torus (examples/scripts/kagome/run_kagome.py in the reference tree; clusters of
lattice_library.py:9-29), same command line and output lines.

    python benchmarking/run_kagome.py 30            # SpinConserve(30, 15) + XParity, eigsolve(nev=2)
    python benchmarking/run_kagome.py 27b --no-z2

Differences that follow from the engine: matrices are always matrix-free (``--shell`` is accepted and implied).
"""
import argparse
import os
import sys
from datetime import datetime

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Solve for the ground state energy of the Heisenberg model on the Kagome lattice.')
    p.add_argument('cluster', nargs='?', default='12', help='which Kagome cluster to use (dynamite_amd.lattices.KAGOME_CLUSTERS)')
    p.add_argument('--shell', action='store_true', help='matrix-free matrices (always on here)')
    p.add_argument('--no-z2', action='store_true', help='do not apply XParity subspace')
    p.add_argument('--nev', type=int, default=2, help='eigenvalues to solve for (the reference script: 2 -- ground state '
                   'and gap; 1 runs Lanczos without a stored basis: four work vectors, the way to the largest clusters)')
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    from dynamite_amd import models
    from dynamite_amd.subspaces import SpinConserve, XParity
    from dynamite_amd.tools import mpi_print
    from dynamite_amd.computations import eigsolve
    mpi_print('Heisenberg interaction on the Kagome lattice')
    mpi_print(f'Cluster: {args.cluster}')
    mpi_print('Use shell matrices: True')
    H = models.kagome(args.cluster)                 # run_kagome.py:20-28
    N = H.get_length()
    subspace = SpinConserve(N, N // 2)              # total magnetization is conserved
    sector = None
    if not args.no_z2:                              # the sector containing the ground state depends on N % 4
        if N % 4 == 0:
            sector = +1
        elif N % 4 == 2:
            sector = -1
    if sector is None:
        mpi_print('Not applying XParity (Z2) subspace')
    else:
        mpi_print(f'XParity (Z2) symmetry sector: {sector}')
        subspace = XParity(subspace, sector=sector)
    mpi_print()
    H.subspace = subspace
    H.shell = True
    tick = datetime.now()
    evals = H.eigsolve(nev=args.nev)
    tock = datetime.now()
    gs_energy = evals[0]
    mpi_print(f'Ground state energy E: {gs_energy}')
    mpi_print(f'E/N: {gs_energy / N}')
    mpi_print()
    if args.nev >= 2:
        gap = evals[1] - gs_energy
        mpi_print(f'Gap: {gap}')
        mpi_print(f'Gap/N: {gap / N}')
        mpi_print()
    mpi_print(f'Solve completed in {tock - tick}')
    st = eigsolve.last_stats or {}
    mpi_print('(%d states, %d multiplies, %s arithmetic, measured residual %.1e, plan: %s)'
              % (subspace.get_dimension(), st.get('matvecs', 0), 'real' if st.get('real_arithmetic') else 'complex128',
                 st.get('max_rel_residual', float('nan')), H.get_mat().describe().strip().split(':')[0]))


if __name__ == '__main__':
    main()
