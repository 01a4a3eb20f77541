"""
Bond graphs of the lattices the reference's example scripts run on.

``kagome(cluster)``: the kagome lattice on the tori of the reference's flagship large-scale example
(``examples/scripts/kagome/lattice_library.py:9-29``: the clusters of Lauchli et al., PRB 83, 212401 (2011) and
PRB 100, 155142 (2019), each given by two spanning vectors in units of the two-site-long cell edge), with the
vertex numbering of ``basis_to_graph`` (``lattice_library.py:32-85``: breadth-first from the origin, neighbours
visited in a fixed order), so that operators built on these edges are the ones the reference script builds
(``run_kagome.py:20-28``).  The reduction of a point into the torus is done here with exact integer arithmetic on the
coordinates in the spanning basis (any representative works: the numbering depends on the graph and the visiting
order only); ``tests/golden/kagome_edges.json`` holds the reference's own edge lists for every cluster.
"""

KAGOME_CLUSTERS = {
    '12': ((2, 0), (0, 2)), '15': ((2, -1), (-1, 3)), '18a': ((2, -1), (0, 3)), '18b': ((2, -2), (-2, -1)),
    '21': ((2, 1), (-1, 3)), '24': ((1, 2), (-3, 2)), '27a': ((2, 1), (-3, 3)), '27b': ((3, 0), (0, 3)),
    '30': ((2, 1), (-2, 4)), '33': ((1, 2), (4, -3)), '36a': ((-2, 3), (4, 0)), '36b': ((3, 0), (-3, 4)),
    '36c': ((3, 0), (-1, 4)), '36d': ((4, -2), (-2, 4)), '39a': ((-1, 3), (5, -2)), '39b': ((1, 3), (-3, 4)),
    '42a': ((-1, 3), (5, -1)), '42b': ((-2, 4), (4, -1)), '48': ((4, 0), (0, 4)),
}

# the six neighbours of a point of the triangular lattice (skew coordinates), in the reference's visiting order
_NEIGHBOURS = ((0, 1), (1, 0), (1, -1), (0, -1), (-1, 0), (-1, 1))


def _is_site(p):
    """The kagome lattice is the triangular lattice without the points (odd, even)."""
    return p[0] % 2 == 0 or p[1] % 2 == 1


def kagome(cluster):
    """(number of sites, sorted list of edges (i, j), i < j) of a kagome torus; ``cluster`` is a name of
    ``KAGOME_CLUSTERS`` or a pair of spanning vectors."""
    (a0, a1), (b0, b1) = KAGOME_CLUSTERS[cluster] if isinstance(cluster, str) else cluster
    A, B = (2 * a0, 2 * a1), (2 * b0, 2 * b1)
    det = A[0] * B[1] - A[1] * B[0]
    if det == 0:
        raise ValueError('spanning vectors are linearly dependent')

    def wrap(p):
        # p = u A + v B; subtract floor(u) A + floor(v) B (u = det(p, B) / det, v = det(A, p) / det)
        nu, nv = p[0] * B[1] - p[1] * B[0], A[0] * p[1] - A[1] * p[0]
        if det < 0:
            nu, nv, d = -nu, -nv, -det
        else:
            d = det
        fu, fv = nu // d, nv // d
        return (p[0] - fu * A[0] - fv * B[0], p[1] - fu * A[1] - fv * B[1])

    index = {(0, 0): 0}
    order = [(0, 0)]
    edges = set()
    at = 0
    while at < len(order):
        x, y = order[at]
        for dx, dy in _NEIGHBOURS:
            q = wrap((x + dx, y + dy))
            if not _is_site(q):
                continue
            j = index.get(q)
            if j is None:
                j = index[q] = len(order)
                order.append(q)
            if at < j:
                edges.add((at, j))
        at += 1
    return len(order), sorted(edges)


def chain(L, periodic=False):
    edges = [(i, i + 1) for i in range(L - 1)]
    if periodic and L > 2:
        edges.append((0, L - 1))
    return L, edges


def square(nx, ny, periodic=True):
    """nx x ny square lattice, site (x, y) -> x + nx * y."""
    edges = set()
    for y in range(ny):
        for x in range(nx):
            i = x + nx * y
            for (u, v) in ((x + 1, y), (x, y + 1)):
                if periodic:
                    u, v = u % nx, v % ny
                elif u >= nx or v >= ny:
                    continue
                j = u + nx * v
                if i != j:
                    edges.add((min(i, j), max(i, j)))
    return nx * ny, sorted(edges)
