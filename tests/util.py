"""Shared helpers for the tests: canonical-limb conversions between Python ints and numpy arrays."""
import numpy as np
import pyoracle as po

CURVES = {0: po.BLS12_381, 1: po.BN254}
FQ_LIMBS = {0: 6, 1: 4}


def limbs(v, n):
    return np.array(po.to_limbs(v, n), dtype=np.uint64)


def fr_arr(vals):
    return np.array([po.to_limbs(v, 4) for v in vals], dtype=np.uint64).reshape(len(vals), 4)


def fr_ints(arr):
    return [po.from_limbs(x) for x in np.asarray(arr).reshape(-1, 4)]


def pt_limbs(curve, group, P):
    """affine point (python) -> flat canonical limbs; None -> zeros"""
    L = FQ_LIMBS[curve]
    if P is None:
        return np.zeros(2 * L * group, dtype=np.uint64)
    if group == 1:
        return np.concatenate([limbs(P[0], L), limbs(P[1], L)])
    return np.concatenate([limbs(P[0][0], L), limbs(P[0][1], L), limbs(P[1][0], L), limbs(P[1][1], L)])


def pts_arr(curve, group, pts):
    return np.stack([pt_limbs(curve, group, P) for P in pts]) if len(pts) else np.zeros((0, 2 * FQ_LIMBS[curve] * group), dtype=np.uint64)


def pt_from_limbs(curve, group, a, inf=0):
    if inf:
        return None
    L = FQ_LIMBS[curve]
    a = np.asarray(a).reshape(-1)
    f = lambda i: po.from_limbs(a[i * L:(i + 1) * L])
    if group == 1:
        return (f(0), f(1))
    return ((f(0), f(1)), (f(2), f(3)))


def jac_to_affine_py(curve, group, jac):
    """Jacobian canonical limbs (3, coord) -> python affine point via big-int inversion (checker side)."""
    C = CURVES[curve]
    G = C.g1 if group == 1 else C.g2
    L = FQ_LIMBS[curve]
    jac = np.asarray(jac).reshape(3, -1)

    def coord(row):
        if group == 1:
            return po.from_limbs(row[:L])
        return (po.from_limbs(row[:L]), po.from_limbs(row[L:2 * L]))

    return G.to_affine((coord(jac[0]), coord(jac[1]), coord(jac[2])))


def group_of(curve, group):
    C = CURVES[curve]
    return C.g1 if group == 1 else C.g2


def qap_domains(zkmod, curve, min_size, two_adicity=None):
    """make_evaluation_domain(min_size) twice: pyoracle's EvaluationDomain and the zkhip_domain describing the same"""
    C = CURVES[curve]
    dom = po.make_evaluation_domain(C, min_size, two_adicity=two_adicity)
    zd = zkmod.zkhip.Domain.make(dom.kind, dom.m, limbs(dom.omega, 4), limbs(dom.shift, 4))
    return dom, zd


def lookup_instance(C, rng, log_n, k_in, k_val, big_inputs=()):
    """A genuine lookup instance in the shape prepare_lookup_value / prepare_lookup_input leave behind: every (theta-compressed) table
    column is zero at row 0, holds distinct non-zero values (one of them repeated in adjacent rows) in the rows after it, zero behind and
    zero from usable_rows on (the mask); every input takes table values (or zero) in the usable rows and anything behind them.  Inputs
    listed in big_inputs live on the 2n-point domain (an expression of degree > 1): f + c (X^n - 1), which reduces to f."""
    r = C.r
    n = 1 << log_n
    usable = n - 3
    values, pool = [], [0]
    for i in range(k_val):
        T = usable // 2 + i
        col = [0] * n
        prev = 0
        for j in range(1, T + 1):
            prev = prev if (j == 3 and prev) else rng.next_mod(r - 1) + 1
            col[j] = prev
            pool.append(prev)
        values.append(col)
    inputs = []
    for i in range(k_in):
        f = [pool[rng.next_mod(len(pool))] for _ in range(usable)] + [rng.next_mod(r) for _ in range(n - usable)]
        if i in big_inputs:
            big = po.dfs_resize(f, 2 * n, C.root_of_unity, r)
            c = rng.next_mod(r)
            f = [(v - 2 * c * (j & 1)) % r for j, v in enumerate(big)]
        inputs.append(f)
    return inputs, values, usable


def permutation_instance(C, rng, log_n, k, usable):
    """A genuine copy-constraint instance in placeholder's shape: the k columns are constant along the cycles of a random permutation of
    the k * usable cells of the usable rows; the rows behind them are blinding (random values, identity permutation), so the grand
    product closes AT usable: V_P[usable] = 1.  -> (columns, S_id, S_sigma)"""
    r = C.r
    n = 1 << log_n
    w, delta = C.root_of_unity(log_n), C.fr_generator
    labels = [[pow(delta, i, r) * pow(w, j, r) % r for j in range(n)] for i in range(k)]
    cells = [(i, j) for i in range(k) for j in range(usable)]
    perm = list(cells)
    for a in range(len(perm) - 1, 0, -1):            # Fisher-Yates with the test's generator
        b = rng.next_mod(a + 1)
        perm[a], perm[b] = perm[b], perm[a]
    sigma = dict(zip(cells, perm))
    val, seen = {}, set()
    for c in cells:
        if c in seen:
            continue
        v, x = rng.next_mod(r), c
        while x not in seen:
            seen.add(x)
            val[x] = v
            x = sigma[x]
    cols = [[val[(i, j)] if j < usable else rng.next_mod(r) for j in range(n)] for i in range(k)]
    S_sigma = [[labels[sigma[(i, j)][0]][sigma[(i, j)][1]] if j < usable else labels[i][j] for j in range(n)] for i in range(k)]
    return cols, labels, S_sigma
