"""
The MSC (mask, sign, coefficient) representation of sums of Pauli strings:
construction helpers for the operators this engine multiplies by.  Semantics
follow ``dynamite.msc_tools`` (reference ``src/dynamite/msc_tools.py``): a term
``(m, s, c)`` is ``c * prod_{i in m} sigma^x_i * prod_{i in s} sigma^z_i`` and
contributes ``(-1)^popcount(bra & s) * c`` to ``H[row, col]`` with
``bra = ket ^ m`` the COLUMN state (msc_tools.py:63-80).
"""
import numpy as np

dnm_int_t = np.int64

# msc_tools.py:14-16
msc_dtype = np.dtype([('masks', dnm_int_t), ('signs', dnm_int_t), ('coeffs', np.complex128)])


def parity(x):
    """Parity of the set bits of (arrays of) non-negative integers (bitwise.py:15-31)."""
    v = np.asarray(x).astype(np.uint64)
    for s in (32, 16, 8, 4, 2, 1):
        v = v ^ (v >> np.uint64(s))
    return (v & np.uint64(1)).astype(np.int64)


def as_msc(x):
    return np.asarray(x, dtype=msc_dtype)


def msc_sum(iterable):
    """Operator addition = concatenation of term lists (msc_tools.py:120-138)."""
    lst = list(iterable)
    if not lst:
        return np.empty(0, dtype=msc_dtype)
    return np.hstack(lst)


def msc_product(iterable):
    """Operator product (msc_tools.py:140-172): for A*B each pair of terms gives
    mask = mA^mB, sign = sA^sB, coeff = cA*cB*(-1)^popcount(mB & sA)."""
    vals = [as_msc(v) for v in iterable]
    rtn = vals[0].copy()
    for term in vals[1:]:
        a = np.repeat(rtn, term.size)
        b = np.tile(term, rtn.size)
        out = np.empty(a.size, dtype=msc_dtype)
        flipped = b['masks'] & a['signs']
        out['masks'] = a['masks'] ^ b['masks']
        out['signs'] = a['signs'] ^ b['signs']
        out['coeffs'] = a['coeffs'] * b['coeffs'] * (1 - 2 * parity(flipped))
        rtn = out
    return rtn


def shift(msc, shift_idx, wrap_idx):
    """Translate along the chain, optionally wrapping at wrap_idx (msc_tools.py:174-223)."""
    if shift_idx == 0:
        return msc
    msc = msc.copy()
    msc['masks'] <<= shift_idx
    msc['signs'] <<= shift_idx
    if wrap_idx is not None:
        hi = (-1) << wrap_idx
        for key in ('masks', 'signs'):
            v = msc[key]
            over = (v & hi) >> wrap_idx
            v |= over
            v &= ~hi
    return msc


def combine_and_sort(msc):
    """Sort by (mask, sign), merge equal terms, drop zeros (msc_tools.py:225-252)."""
    msc = as_msc(msc)
    if msc.size == 0:
        return msc.copy()
    order = np.lexsort((msc['signs'], msc['masks']))
    s = msc[order]
    new = np.ones(s.size, dtype=bool)
    new[1:] = (s['masks'][1:] != s['masks'][:-1]) | (s['signs'][1:] != s['signs'][:-1])
    starts = np.nonzero(new)[0]
    out = np.empty(starts.size, dtype=msc_dtype)
    out['masks'] = s['masks'][starts]
    out['signs'] = s['signs'][starts]
    # sum in original input order within each group, as the reference's loop does
    group = np.cumsum(new) - 1
    inv = np.empty(msc.size, dtype=np.int64)
    inv[order] = group
    coeffs = np.zeros(starts.size, dtype=np.complex128)
    np.add.at(coeffs, inv, msc['coeffs'])
    out['coeffs'] = coeffs
    return out[out['coeffs'] != 0]


def is_hermitian(msc):
    """msc_tools.py:94-118: a term is imaginary iff parity(mask & sign) is odd."""
    odd = parity(msc['masks'] & msc['signs']) == 1
    if np.any(np.real(msc['coeffs'][odd])):
        return False
    if np.any(np.imag(msc['coeffs'][~odd])):
        return False
    return True


def truncate(msc, tol):
    if tol < 0:
        raise ValueError('tol cannot be less than zero')
    return msc[np.abs(msc['coeffs']) > tol]


def max_spin_idx(msc):
    """Largest spin index the operator touches, -1 if empty (msc_tools.py:367-386)."""
    if msc.size == 0:
        return -1
    top = int(max(np.max(msc['masks']), np.max(msc['signs'])))
    return top.bit_length() - 1


def nnz(msc):
    return len(np.unique(msc['masks']))


def get_mask_offsets(msc):
    """Unique masks and the index where each one's terms start, plus the end
    sentinel (Operator._get_mask_offsets, operators.py:653-669)."""
    if not np.all(np.diff(msc['masks']) >= 0):
        raise ValueError('msc must be sorted first')
    masks, indices = np.unique(msc['masks'], return_index=True)
    offsets = np.empty(indices.size + 1, dtype=dnm_int_t)
    offsets[:-1] = indices
    offsets[-1] = msc.shape[0]
    return masks.astype(dnm_int_t), offsets


def serialize(msc):
    """``nterms\\nint_size\\n`` + big-endian masks, signs, coeffs (msc_tools.py:276-311)."""
    rtn = (str(msc.size) + '\n').encode('utf-8')
    rtn += (str(msc.dtype['masks'].itemsize * 8) + '\n').encode('utf-8')
    rtn += msc['masks'].astype('>i8').tobytes()
    rtn += msc['signs'].astype('>i8').tobytes()
    rtn += msc['coeffs'].astype('>c16').tobytes()
    return rtn


def deserialize(data):
    """Inverse of serialize (msc_tools.py:313-365); accepts 32- and 64-bit files."""
    stop = data.find(b'\n')
    n = int(data[:stop])
    start = stop + 1
    stop = data.find(b'\n', start)
    int_size = int(data[start:stop])
    if int_size not in (32, 64):
        raise ValueError('Invalid int_size. Perhaps file is corrupt.')
    it = np.dtype('>i4' if int_size == 32 else '>i8')
    start = stop + 1
    nb = n * int_size // 8
    msc = np.empty(n, dtype=msc_dtype)
    msc['masks'] = np.frombuffer(data[start:start + nb], dtype=it)
    msc['signs'] = np.frombuffer(data[start + nb:start + 2 * nb], dtype=it)
    msc['coeffs'] = np.frombuffer(data[start + 2 * nb:], dtype='>c16')
    return msc


def msc_to_numpy(msc, dims, idx_to_state=None, state_to_idx=None, sparse=True):
    """Explicit matrix of an MSC operator (host side; used by Operator.to_numpy
    for small systems).  Definition: msc_tools.py:19-92."""
    import scipy.sparse
    msc = as_msc(msc)
    M, N = dims
    rows = np.arange(M, dtype=dnm_int_t)
    kets = rows if idx_to_state is None else np.asarray(idx_to_state(rows), dtype=dnm_int_t)
    data, ri, ci = [], [], []
    for m, s, c in msc:
        bra = kets ^ m
        col = bra if state_to_idx is None else np.asarray(state_to_idx(bra), dtype=dnm_int_t)
        good = col != -1
        data.append((1 - 2 * parity(bra[good] & s)) * c)
        ri.append(rows[good])
        ci.append(col[good])
    if data:
        data, ri, ci = np.concatenate(data), np.concatenate(ri), np.concatenate(ci)
    else:
        data, ri, ci = np.zeros(0, complex), np.zeros(0, dnm_int_t), np.zeros(0, dnm_int_t)
    ary = scipy.sparse.csc_matrix((data, (ri, ci)), shape=dims)
    return ary if sparse else ary.toarray()
