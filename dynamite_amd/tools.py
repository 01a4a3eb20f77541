"""
Small helpers dynamite scripts use around the hot path (reference ``src/dynamite/tools.py``):
rank-aware printing, version information, and memory accounting -- here the device (HBM) memory
of the process's GPU, which is where states and Krylov bases live.
"""
import warnings

from .config import config


class _CommWorld:
    """What scripts use of ``MPI_COMM_WORLD()`` (tools.py:8-16): rank, size, barrier -- over
    torch.distributed here."""

    @property
    def rank(self):
        return config.rank

    @property
    def size(self):
        return config.world_size

    def barrier(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()


def MPI_COMM_WORLD():
    return _CommWorld()


def mpi_print(*args, rank=0, **kwargs):
    """``print`` from a single rank only (tools.py:19-27); ranks are torch.distributed ranks."""
    if config.rank == rank:
        print(*args, **kwargs)


def complex_enabled():
    """This engine is always complex128 (tools.py:187-191)."""
    return True


def get_version():
    """Version information: the engine's C-ABI version, ROCm/HIP and PyTorch (tools.py:30-56 reports
    dynamite/PETSc/SLEPc; there is no PETSc or SLEPc here)."""
    import torch
    from . import _lib
    return {'dynamite_amd': {'abi': int(_lib.lib().dnm_version())},
            'torch': torch.__version__, 'hip': getattr(torch.version, 'hip', None)}


def get_version_str():
    v = get_version()
    return 'dynamite_amd ABI %d; torch %s; HIP %s' % (v['dynamite_amd']['abi'], v['torch'], v['hip'])


def track_memory():
    """Start tracking the peak for ``get_memory_usage(max_usage=True)`` (tools.py:86-94)."""
    import torch
    config._initialize()
    torch.cuda.reset_peak_memory_stats()


def get_memory_usage(group_by='all', max_usage=False):
    """Device memory held by this engine's allocations, in gigabytes (tools.py:96-153).
    ``group_by``: 'rank' (this process), 'node' or 'all' (summed over the ranks; one node here)."""
    import torch
    config._initialize()
    if group_by not in ('rank', 'node', 'all'):
        raise ValueError(f"group_by must be 'rank', 'node', or 'all'; got '{group_by}'")
    free, total = torch.cuda.mem_get_info()
    # device memory in use outside torch's caching allocator (the engine's tables and cached Krylov
    # workspace) plus what torch has handed out (states, vectors)
    outside = max(0, (total - free) - torch.cuda.memory_reserved())
    local = outside + (torch.cuda.max_memory_allocated() if max_usage else torch.cuda.memory_allocated())
    local /= 1E9
    if group_by == 'rank' or config.world_size == 1:
        return local
    import torch.distributed as dist
    t = torch.tensor([local], dtype=torch.float64, device=config.device)
    dist.all_reduce(t)
    return float(t.item())


def get_max_memory_usage(which='all'):
    """[deprecated, tools.py:155-169]"""
    if which != 'all':
        raise ValueError('values of "which" other than "all" no longer supported')
    warnings.warn("get_max_memory_usage() is deprecated; use get_memory_usage(max_usage=True) instead",
                  DeprecationWarning, stacklevel=2)
    return get_memory_usage(group_by='rank', max_usage=True)


def get_cur_memory_usage(which='all'):
    """[deprecated, tools.py:171-185]"""
    if which != 'all':
        raise ValueError('values of "which" other than "all" no longer supported')
    warnings.warn("get_cur_memory_usage() is deprecated; use get_memory_usage() instead",
                  DeprecationWarning, stacklevel=2)
    return get_memory_usage(group_by='rank')
