#!/usr/bin/env python
"""
The reference's benchmark harness (benchmarking/benchmark.py) on this engine: the same command line and the
same '---RESULTS---' key,value lines, so existing sweep scripts and parsers keep working.

    python benchmarking/benchmark.py -L 26 -H MBL --shell --mult --mult_count 10
    python benchmarking/benchmark.py -L 24 -H heisenberg --subspace spinconserve --eigsolve --nev 1
    python -m torch.distributed.run --nproc-per-node 8 benchmarking/benchmark.py -L 33 -H MBL --mult

Differences that follow from the engine: matrices are always matrix-free (``--shell`` is accepted and implied),
``--gpu`` is implied, ``--slepc_args`` is ignored, timings are taken with the device synchronised.
"""
import argparse
import os
import sys
from itertools import combinations
from random import seed, uniform
from timeit import default_timer

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HAMILTONIANS = ('MBL', 'long_range', 'SYK', 'ising', 'XX', 'heisenberg')


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Benchmark dynamite_amd.')
    p.add_argument('-L', type=int, required=True, help='spin chain length')
    p.add_argument('-H', choices=HAMILTONIANS, default=None, help='Hamiltonian')
    p.add_argument('--shell', action='store_true', help='matrix-free matrices (always on here)')
    p.add_argument('--no-precompute-diagonal', action='store_true')
    p.add_argument('--gpu', action='store_true', help='run on the GPU (always on here)')
    p.add_argument('--slepc_args', type=str, default='', help='ignored')
    p.add_argument('--subspace', choices=['full', 'parity', 'spinconserve', 'auto', 'nosortauto'], default='full')
    p.add_argument('--which_space', type=str, help='parity sector / number of down spins / Auto start state')
    p.add_argument('--xparity', choices=['plus', 'minus'], nargs='?', const='plus')
    p.add_argument('--evolve', action='store_true')
    p.add_argument('-t', type=float, default=50.0, help='evolution time (in units of 1/||H|| unless --no_normalize_t)')
    p.add_argument('--no_normalize_t', action='store_true')
    p.add_argument('--mult', action='store_true')
    p.add_argument('--mult_count', type=int, default=1)
    p.add_argument('--norm', action='store_true')
    p.add_argument('--eigsolve', action='store_true')
    p.add_argument('--nev', type=int, default=1)
    p.add_argument('--target', type=float, help='not supported for matrix-free operators (as in the reference)')
    p.add_argument('--rdm', action='store_true')
    p.add_argument('--keep', type=lambda s: [int(x) for x in s.split(',')])
    p.add_argument('--check-conserves', action='store_true')
    a = p.parse_args(argv)
    if a.evolve and not a.no_normalize_t:
        a.norm = True           # needed to scale t; benchmarked on the way
    return a


def build_hamiltonian(a):
    from dynamite_amd.operators import sigmax, sigmay, sigmaz, index_sum, op_sum, op_product
    from dynamite_amd.extras import majorana
    L = a.L
    heis = lambda: index_sum(op_sum(0.25 * s(0) * s(1) for s in (sigmax, sigmay, sigmaz)))      # noqa: E731
    if a.H == 'MBL':
        H = heis()
        seed(0)
        for i in range(L):
            H += uniform(-3, 3) * 0.5 * sigmaz(i)
    elif a.H == 'heisenberg':
        H = heis()
    elif a.H == 'XX':
        H = index_sum(0.25 * sigmax(0) * sigmax(1))
    elif a.H == 'ising':
        H = index_sum(0.25 * sigmaz(0) * sigmaz(1)) + 0.1 * index_sum(sigmax())
    elif a.H == 'long_range':
        H = op_sum(index_sum(0.25 * sigmaz(0) * sigmaz(i)) for i in range(1, L))
        H += 0.5 * index_sum(0.25 * sigmax(0) * sigmax(1))
        H += sum(0.05 * index_sum(s()) for s in (sigmax, sigmay, sigmaz))
    elif a.H == 'SYK':
        seed(0)
        maj = [majorana(i) for i in range(2 * L)]

        def products():
            for idxs in combinations(range(2 * L), 4):
                term = op_product(maj[i] for i in idxs)
                term.scale(uniform(-1, 1))
                yield term
        H = op_sum(products())
        H.scale(np.sqrt(6 / (2 * L) ** 3))
    else:
        raise ValueError('Unrecognized Hamiltonian.')
    H.allow_projection = True        # the conservation check is benchmarked separately
    return H


def build_subspace(a, H=None):
    from dynamite_amd.subspaces import Full, Parity, SpinConserve, Auto, XParity
    w = a.which_space
    if a.subspace == 'full':
        sub = Full()
    elif a.subspace == 'parity':
        sub = Parity(w if w is not None else 'even')
    elif a.subspace == 'spinconserve':
        sub = SpinConserve(a.L, int(w) if w is not None else a.L // 2)
    elif a.subspace in ('auto', 'nosortauto'):
        start = w if w is not None else 'U' * (a.L // 2) + 'D' * (a.L - a.L // 2)
        sub = Auto(H, start, sort=a.subspace == 'auto')
    else:
        raise ValueError('invalid subspace')
    if a.xparity is not None:
        sub = XParity(sub, sector={'plus': '+', 'minus': '-'}[a.xparity])
    return sub


def main():
    t_start = default_timer()
    a = parse_args()
    import torch
    from dynamite_amd import config
    from dynamite_amd.states import State
    from dynamite_amd.computations import reduced_density_matrix
    from dynamite_amd.tools import mpi_print, track_memory, get_memory_usage
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    config.initialize(a.slepc_args.split(' '), gpu=True)
    config.L = a.L
    mpi_print('---ARGUMENTS---')
    for k, v in vars(a).items():
        mpi_print('%s,%s' % (k, v))
    track_memory()
    stats = {}

    def timed(name, fn, *args, **kw):
        torch.cuda.synchronize()
        t0 = default_timer()
        out = fn(*args, **kw)
        torch.cuda.synchronize()
        stats[name] = default_timer() - t0
        return out

    H = None
    if a.H is not None:
        H = timed('build_hamiltonian', build_hamiltonian, a)
    elif a.subspace in ('auto', 'nosortauto') or a.norm or a.eigsolve or a.evolve or a.mult:
        raise ValueError('Must specify Hamiltonian for this benchmark.')
    sub = timed('build_subspace', build_subspace, a, H)
    if sub.L is None:
        sub.L = a.L
    if H is not None:
        H.subspace = sub
        if a.no_precompute_diagonal:
            H.precompute_diagonal = False
        mpi_print('H statistics:')
        mpi_print(' dim:', H.dim[0])
        mpi_print(' nnz:', H.nnz)
        mpi_print(' density:', H.density)
        mpi_print(' nterms:', H.nterms)
        timed('build_mat', H.build_mat)
    x = y = None
    if a.evolve or a.mult or a.rdm:
        x, y = State(L=a.L, subspace=sub), State(L=a.L, subspace=sub)
        # (a benchmark state needs no particular stream: from 2^22 amplitudes on the device's generator fills it --
        # the host route that reproduces the reference's numpy stream takes 1.7 s at L=26)
        timed('set_random_state', x.set_random, device_rng=True if sub.get_dimension() >= (1 << 22) else None)
    if a.norm:
        timed('compute_norm', H.infinity_norm)
    if a.eigsolve:
        timed('do_eigsolve', H.eigsolve, nev=a.nev, target=a.target)
    if a.evolve:
        t = a.t if a.no_normalize_t else a.t / H.infinity_norm()
        timed('do_evolve', H.evolve, x, t=t, result=y)
    if a.mult:
        def mults():
            for _ in range(a.mult_count):
                H.dot(x, y)
        timed('do_mult', mults)
        stats['avg_mult_time'] = stats['do_mult'] / a.mult_count
    if a.rdm:
        timed('do_rdm', reduced_density_matrix, x, a.keep if a.keep is not None else list(range(a.L // 2)))
    if a.check_conserves:
        timed('do_check_conserves', H.conserves, H.subspace)
    stats['Gb_memory'] = get_memory_usage(group_by='all', max_usage=True)
    if H is not None:
        H.destroy_mat()
    stats['total_time'] = default_timer() - t_start
    mpi_print('---RESULTS---')
    for k, v in stats.items():
        mpi_print('{0}, {1:0.4f}'.format(k, v))


if __name__ == '__main__':
    main()
