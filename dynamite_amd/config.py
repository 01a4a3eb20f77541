"""
Global configuration, the minimal equivalent of ``dynamite.config``
(reference ``src/dynamite/__init__.py:12-227``): default ``L``, default
``subspace``, the shell flag (this engine is matrix-free only) and the device /
rank this process drives.  One process per GPU; ranks come from
``torch.distributed`` when it is initialised (backend "nccl" = RCCL on ROCm,
"gloo" in CPU tests).
"""
import os


_warned = set()


def knob(name, default=None):
    """Experiment knobs (DNM_* environment variables) count only under DNM_EXPERIMENTAL=1, as in the native
    library (csrc/dnm_common.h: knob); tests and the A/B tools set the gate, production runs ignore them -- and say
    so once per knob, so that an A/B run without the gate does not silently measure the default twice."""
    if os.environ.get('DNM_EXPERIMENTAL') != '1':
        if os.environ.get(name) and name not in _warned:
            _warned.add(name)
            import warnings
            warnings.warn('%s is set but ignored: experiment knobs need DNM_EXPERIMENTAL=1' % name, stacklevel=2)
        return default
    v = os.environ.get(name)
    return v if v else default


class _Config:
    def __init__(self):
        self._L = None
        self._subspace = None
        self._shell = True
        self._initialized = False
        self.gpu = True
        self.device = None           # torch.device of this rank
        # layout of Full / Parity state vectors in device memory: XOR-swizzle shift (include/dynamite_amd.h,
        # dnm_subspace.vec_swizzle); 0 = index order.  16 measured best on MI355X (profiles/r02_exp1_swz.txt).
        # Fixed for the life of the process: vectors and matrices built under different values do not mix.
        self.vec_swizzle = int(knob('DNM_SWZ', '16'))
        # layout of SpinConserve state vectors on one rank: (a, w) = low bits / window bits of the three-field
        # internal layout (csrc/sc3.h) the two-pass SpinConserve multiply works in, or None for reference order;
        # used from sc_layout_min_dim states on (smaller subspaces keep reference order and the row kernels).
        # (14, 10): 55 KB and 63 KB LDS tiles, two workgroups per CU (profiles/r03_exp3_sc3_v2.txt).
        lay = knob('DNM_SC_LAYOUT', '14,10')
        self.sc_layout = None if lay in ('0', '') else tuple(int(v) for v in lay.split(','))
        self.sc_layout_min_dim = int(knob('DNM_SC_LAYOUT_MIN_DIM', str(1 << 22)))
        # operators on a bond graph in such a subspace (one rank): relabel the spins so that as many pair hops as the
        # graph allows fall inside the layout's fields (backend._relabelled, csrc/sc3_perm.cpp); chains keep the identity
        self.sc_site_perm = knob('DNM_SC_SITE_PERM', '1') != '0'
        # XParity on top of a SpinConserve subspace in the internal layout: its vectors are the layout's first half
        # (one rank); DNM_SC_XPARITY_LAYOUT=0 keeps them in reference order (the row kernels)
        self.sc_xparity_layout = knob('DNM_SC_XPARITY_LAYOUT', '1') != '0'
        # partitioned multiplies as ONE native call (dnm_mat_mult_partitioned, csrc/comm.cpp: the library's own RCCL
        # communicator and exchange stream; the reference posts its scatters inside C as well,
        # bpetsc_template_2.c:413-504) instead of the host schedules over torch.distributed.
        # None (the default): whenever the process group's transport is RCCL (backend "nccl") -- the host schedules
        # remain for gloo-staged ranks (CPU tests, several ranks on one GPU); True: always (tests: a stand-in transport
        # named by DNM_RCCL_LIB under a gloo process group); False: never.  DNM_NATIVE_COMM=0 / 1 sets it from outside --
        # a production switch, not an experiment knob: the way back to the host schedules should the native one
        # misbehave on a machine nobody has tested it on.
        v = os.environ.get('DNM_NATIVE_COMM', '')
        self.native_comm = None if v == '' else v == '1'
        # evolve / eigsolve of a partitioned SpinConserve operator in the internal layout iterate on the partition made for
        # the exchange (Operator.get_solver_mat; states move there and back block by block, backend.reorder_blocks;
        # DNM_SC_SOLVER_PARTITION=0: everything on the reference-compatible one)
        self.sc_solver_partition = knob('DNM_SC_SOLVER_PARTITION', '1') != '0'
        # eigsolve of a real-symmetric operator (every matrix element real in the product basis): real arithmetic on
        # vectors stored two amplitudes to a complex128 element (Full / Parity, on a power-of-two number of ranks) or one
        # double per position of the internal layout (SpinConserve, any rank count) -- DNM_MAT_REAL_PACKED, half the
        # bytes per multiply and per Krylov vector; eigenvectors are handed back as complex states.
        # None: from 2^23 amplitudes per rank on (the smallest rank decides); True / False: always / never.
        self.eigs_real_arithmetic = None

    # -- L / subspace / shell: same validation as the reference --------------
    @property
    def L(self):
        return self._L

    @L.setter
    def L(self, value):
        if value is not None:
            if int(value) != value or value < 0 or value > 63:
                raise ValueError('L must be an integer in [0, 63]')
            value = int(value)
        self._L = value

    @property
    def subspace(self):
        return self._subspace

    @subspace.setter
    def subspace(self, value):
        from .subspaces import Subspace
        if value is not None and not isinstance(value, Subspace):
            raise ValueError('subspace can only be set to objects of Subspace type')
        self._subspace = value

    @property
    def shell(self):
        return self._shell

    @shell.setter
    def shell(self, value):
        if not isinstance(value, bool):
            raise ValueError('Shell must be set to True or False.')
        if not value:
            import warnings
            warnings.warn('dynamite_amd is matrix-free: shell=False is accepted and ignored', stacklevel=2)
        self._shell = True

    # -- process / device -------------------------------------------------------
    @property
    def rank(self):
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    @property
    def world_size(self):
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def initialize(self, slepc_args=None, version_check=False, gpu=True):
        """Once-only initialisation (``config.initialize``, __init__.py:24-49).
        ``slepc_args`` is accepted for call compatibility and ignored."""
        if self._initialized:
            raise RuntimeError('initialize has already been called.')
        self._initialize(gpu=gpu)

    def _initialize(self, gpu=True):
        if self._initialized:
            return
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('dynamite_amd needs an AMD GPU (torch.cuda.is_available() is False); '
                               'there is no CPU fallback')
        local = int(os.environ.get('LOCAL_RANK', '0'))
        self.device = torch.device('cuda', local % max(1, torch.cuda.device_count()))
        torch.cuda.set_device(self.device)
        from . import _lib
        _lib.check(_lib.lib().dnm_set_device(self.device.index))
        self._initialized = True

    @property
    def initialized(self):
        return self._initialized


config = _Config()
