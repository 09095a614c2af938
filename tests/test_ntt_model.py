"""Line-by-line CPU replay of the index arithmetic of crypto3-zk_amd/csrc/ntt.hip (pass planning, tile
bit-reversed tile load, in-LDS DIT stages fused two per round trip, store with the next pass's twiddle (what
ntt_build_tw tabulates) and both store decodings,
ping-pong buffer choice) on Python integers, checked against the O(n^2) DFT.  Catches indexing mistakes
without a GPU; the arithmetic itself is covered by test_host_arith.py."""
import random

import pytest

import pyoracle as po


def plan(log_m, smax=8, tile_log=3):
    np_ = (log_m + smax - 1) // smax
    sv = [log_m // np_ + (1 if i < log_m % np_ else 0) for i in range(np_)]
    passes = []
    log_ns = 0
    for s in sv:
        log_cols = log_m - s
        log_t = min(tile_log, log_cols)
        while s + log_t > 12 and log_t > 0:
            log_t -= 1
        passes.append((s, log_ns, log_t))
        log_ns += s
    return passes


def bitrev(v, bits):
    return int(bin(v)[2:].zfill(bits)[::-1], 2) if bits else 0


def kernel_pass(x, log_m, s, log_ns, log_t, next_s, w, r, pre=None, post=None, scale=1):
    m = 1 << log_m
    R, T = 1 << s, 1 << log_t
    nelem = R * T
    log_stride = log_m - s
    ns_mask = (1 << log_ns) - 1
    last = next_s == 0
    n_log_ns = log_ns + s
    n_log_stride = log_m - next_s
    n_ns_mask = (1 << n_log_ns) - 1
    n_tw_shift = log_m - n_log_ns - next_s
    y = [None] * m
    for tile in range(1 << (log_m - s - log_t)):
        j0 = tile << log_t
        twr = [pow(w, q << log_stride, r) for q in range(R // 2)]
        lds = [0] * nelem
        for e in range(nelem):
            t, c = e >> log_t, e & (T - 1)
            gi = j0 + c + (t << log_stride)
            v = x[gi]
            if pre is not None:
                v = v * pow(pre, gi, r) % r
            lds[(bitrev(t, s) << log_t) + c] = v
        # the kernel runs stage 0 alone when s is odd, then TWO stages per LDS round trip: a thread holds the rows
        # i0 + {0, h, 2h, 3h} (ntt_pass: "two per round trip"); replayed here in exactly that order
        st = 0
        if s & 1:
            for bf in range(nelem >> 1):
                c, q = bf & (T - 1), bf >> log_t
                e0 = ((q << 1) << log_t) + c
                e1 = e0 + T
                a, b = lds[e0], lds[e1]
                lds[e0], lds[e1] = (a + b) % r, (a - b) % r
            st = 1
        while st < s:
            h = 1 << st
            for gq in range(nelem >> 2):
                c, q = gq & (T - 1), gq >> log_t
                qq = q & (h - 1)
                i0 = ((q - qq) << 2) + qq
                e = [((i0 + k * h) << log_t) + c for k in range(4)]
                x0, x1, x2, x3 = (lds[i] for i in e)
                w1 = twr[qq << (s - 1 - st)] if st != 0 else 1
                w2a, w2b = twr[qq << (s - 2 - st)], twr[(qq + h) << (s - 2 - st)]
                x1, x3 = x1 * w1 % r, x3 * w1 % r
                a0, a1, a2, a3 = (x0 + x1) % r, (x0 - x1) % r, (x2 + x3) % r, (x2 - x3) % r
                assert st != 0 or w2a == 1  # the kernel skips this product in the first round
                b2, b3 = a2 * w2a % r, a3 * w2b % r
                lds[e[0]], lds[e[2]], lds[e[1]], lds[e[3]] = (a0 + b2) % r, (a0 - b2) % r, (a1 + b3) % r, (a1 - b3) % r
            st += 2
        for e in range(nelem):
            if log_ns >= log_t:
                tp, c = e >> log_t, e & (T - 1)
            else:
                k_lo = e & ns_mask
                tp = (e >> log_ns) & (R - 1)
                c_hi = e >> (log_ns + s)
                c = (c_hi << log_ns) + k_lo
            j = j0 + c
            k = j & ns_mask
            oi = ((j - k) << s) + k + (tp << log_ns)
            v = lds[(tp << log_t) + c]
            if not last:
                jn, tn = oi & ((1 << n_log_stride) - 1), oi >> n_log_stride
                ex = ((jn & n_ns_mask) * tn) << n_tw_shift
                assert ex < m
                f = pow(w, ex, r)
            else:
                f = scale
                if post is not None:
                    f = f * pow(post, oi, r) % r
            assert y[oi] is None
            y[oi] = v * f % r
    assert all(v is not None for v in y)
    return y


def model_ntt(a, log_m, w, r, inverse=False, coset=None, smax=8, tile_log=3):
    if log_m == 0:
        return list(a)
    weff = pow(w, -1, r) if inverse else w
    geff = None if coset is None else (pow(coset, -1, r) if inverse else coset)
    passes = plan(log_m, smax, tile_log)
    x = list(a)
    for i, (s, log_ns, log_t) in enumerate(passes):
        last = i == len(passes) - 1
        next_s = 0 if last else passes[i + 1][0]
        x = kernel_pass(x, log_m, s, log_ns, log_t, next_s, weff, r,
                        pre=geff if (not inverse and coset is not None and i == 0) else None,
                        post=geff if (inverse and coset is not None and last) else None,
                        scale=pow(1 << log_m, -1, r) if inverse else 1)
    return x


@pytest.mark.parametrize("log_m,smax,tile_log", [(1, 8, 3), (2, 8, 3), (3, 2, 1), (4, 2, 3), (5, 2, 1), (5, 3, 3), (6, 2, 2),
                                                 (6, 3, 0), (7, 3, 2), (7, 4, 3), (8, 3, 1), (9, 4, 2)])
def test_kernel_index_model(log_m, smax, tile_log):
    C = po.BLS12_381
    r = C.r
    random.seed(log_m * 100 + smax)
    m = 1 << log_m
    w = C.root_of_unity(log_m)
    a = [random.randrange(r) for _ in range(m)]
    d = po.ntt(a, w, r)
    if m <= 64:
        assert d == po.dft_naive(a, w, r)
    assert model_ntt(a, log_m, w, r, smax=smax, tile_log=tile_log) == d
    assert model_ntt(d, log_m, w, r, inverse=True, smax=smax, tile_log=tile_log) == a
    g = C.fr_generator
    dc = po.ntt(po.multiply_by_coset(a, g, r), w, r)
    assert model_ntt(a, log_m, w, r, coset=g, smax=smax, tile_log=tile_log) == dc
    assert model_ntt(dc, log_m, w, r, inverse=True, coset=g, smax=smax, tile_log=tile_log) == a


def test_buffer_schedule():
    """Mirror of the buffer choice in ntt_run_t: the first pass reads and the last pass writes the caller's vector, the
    intermediates alternate between two workspace buffers (limb form), no pass reads what it writes unless it is alone."""
    for np_ in range(1, 9):
        src = "data"
        for i in range(np_):
            dst = "data" if i == np_ - 1 else ("A", "B")[i & 1]
            if np_ > 1:
                assert dst != src
            src = dst
        assert src == "data"
