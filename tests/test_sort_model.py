"""Host-side MODEL of the long-list tile sort's index arithmetic (splatter360_amd/csrc/s360_forward.hip:
k_tile_scan's chunk table, chunk_unit, k_sort_chunks, merge_path_global_wave, k_merge_pass): the same formulas,
executed in Python on random unique 64-bit keys, for list lengths the GPU parity tests do not hit exactly (exact
multiples of the chunk size, one key over, the full pass budget).  It checks the arithmetic — run boundaries,
ping-pong parity (the last pass must land in `keys`), merge-path partitions with an unpaired tail run — not the
kernels themselves (those are compared with the oracle in tests/test_gpu_edge_cases.py)."""
import numpy as np
import pytest

SORT_SHORT, SORT_CHUNK, MAX_PASSES, E, THREADS = 2048, 4096, 4, 8, 512
LDSM_CHUNKS, LDSM_E = 4, 32          # lists of 2..4 chunks: ONE workgroup merges the whole list in LDS (merge_tile_lds)


def ceil_log2(x):
    return 0 if x <= 1 else int(x - 1).bit_length()


def merge_path_wave(A, la, B, lb, d):
    """64-ary search of merge_path_global_wave: number of A elements among the first d merged outputs."""
    lo, hi = max(d - lb, 0), min(d, la)
    while lo < hi:
        step = (hi - lo + 63) // 64
        cnt = 0
        for lane in range(64):
            a = lo + lane * step
            if a < hi and A[a] <= B[d - 1 - a]:
                cnt += 1                      # the predicate holds for a prefix of the probes
        if cnt == 0:
            hi = lo
        else:
            nhi = lo + cnt * step
            lo = lo + (cnt - 1) * step + 1
            hi = min(nhi, hi)
    return lo


def merge_passes_of(nch):
    return 0 if nch <= 1 else 1 if nch <= LDSM_CHUNKS else ceil_log2(nch)


def merge_tile_lds_model(src, n):
    """merge_tile_lds: in-LDS merge of the 4 096-key sorted runs of a list of n <= 16 384 keys, ragged last run included; every
    thread produces LDSM_E consecutive outputs per pass from a merge-path binary search."""
    INF = np.uint64(0xFFFFFFFFFFFFFFFF)
    lds = np.array(src[:n], dtype=np.uint64)
    width = SORT_CHUNK
    while width < n:
        out = lds.copy()
        for t in range(THREADS):
            out0 = t * LDSM_E
            if out0 >= n:
                break
            pa = out0 // (2 * width) * (2 * width)
            pb = pa + width
            la, lb = min(width, n - pa), (min(width, n - pb) if pb < n else 0)
            diag = out0 - pa
            lo, hi = max(diag - lb, 0), min(diag, la)
            while lo < hi:
                mid = (lo + hi) >> 1
                if lds[pa + mid] <= lds[pb + diag - 1 - mid]:
                    lo = mid + 1
                else:
                    hi = mid
            a, b = lo, diag - lo
            for q in range(LDSM_E):
                ka = lds[pa + a] if a < la else INF
                kb = lds[pb + b] if b < lb else INF
                take_a = ka <= kb
                if out0 + q < n:
                    out[out0 + q] = ka if take_a else kb
                a, b = (a + 1, b) if take_a else (a, b + 1)
        lds = out
        width <<= 1
    return lds


def sort_tile_model(keys_in):
    n = len(keys_in)
    assert n > SORT_SHORT
    nch = (n + SORT_CHUNK - 1) // SORT_CHUNK
    passes = merge_passes_of(nch)
    assert passes <= MAX_PASSES
    bufs = [np.array(keys_in, dtype=np.uint64), np.zeros(n, np.uint64)]     # 0 = keys, 1 = alt
    lst = np.zeros(n, np.uint32)
    src0 = bufs[0].copy()
    for k in range(nch):                                                    # k_sort_chunks
        c0, ln = k * SORT_CHUNK, min(SORT_CHUNK, n - k * SORT_CHUNK)
        bufs[passes & 1][c0:c0 + ln] = np.sort(src0[c0:c0 + ln])
        if passes == 0:
            lst[c0:c0 + ln] = (bufs[0][c0:c0 + ln] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    if 2 <= nch <= LDSM_CHUNKS:                                             # k_merge_all, whole-list unit: alt -> keys, one write step
        assert passes == 1
        res = merge_tile_lds_model(bufs[1], n)
        bufs[0][:] = res
        return bufs[0], (res & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    for p in range(passes):                                                 # k_merge_all, (pass, chunk) units
        R = SORT_CHUNK << p
        src, dst = bufs[(passes - p) & 1], bufs[(passes - p - 1) & 1]
        out = dst.copy()
        for k in range(nch):
            o_tile, ln = k * SORT_CHUNK, min(SORT_CHUNK, n - k * SORT_CHUNK)
            pair0 = o_tile // (2 * R) * (2 * R)
            la = min(R, n - pair0)
            lb = min(R, n - pair0 - la)
            A, B = src[pair0:pair0 + la], src[pair0 + la:pair0 + la + lb]
            o = o_tile - pair0
            a0, a1 = merge_path_wave(A, la, B, lb, o), merge_path_wave(A, la, B, lb, o + ln)
            b0, b1 = o - a0, o + ln - a1
            na, nb = a1 - a0, b1 - b0
            assert na + nb == ln and 0 <= na and 0 <= nb
            lds = np.concatenate([A[a0:a1], B[b0:b1]])
            res = np.zeros(ln, np.uint64)
            for t in range(THREADS):                                        # per-thread merge path + 8-step serial merge
                out0 = t * E
                if out0 >= ln:
                    break
                lo_a, hi_a = max(out0 - nb, 0), min(out0, na)
                while lo_a < hi_a:
                    mid = (lo_a + hi_a) >> 1
                    if lds[mid] <= lds[na + out0 - 1 - mid]:
                        lo_a = mid + 1
                    else:
                        hi_a = mid
                a, b = lo_a, out0 - lo_a
                for q in range(E):
                    ka = lds[a] if a < na else np.uint64(0xFFFFFFFFFFFFFFFF)
                    kb = lds[na + b] if b < nb else np.uint64(0xFFFFFFFFFFFFFFFF)
                    take_a = ka <= kb
                    if out0 + q < ln:
                        res[out0 + q] = ka if take_a else kb
                    a, b = (a + 1, b) if take_a else (a, b + 1)
            out[o_tile:o_tile + ln] = res
            if p + 1 == passes:
                lst[o_tile:o_tile + ln] = (res & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        dst[:] = out
    return bufs[0], lst


@pytest.mark.parametrize("n", [2049, 4096, 4097, 8192, 8193, 9000, 12288, 12289, 13751, 16383, 16384, 16385, 20001, 32768, 32769, 65535, 65536])
def test_chunk_and_merge_pass_arithmetic(n):
    rng = np.random.default_rng(n)
    depth = rng.integers(0, 1 << 20, n).astype(np.uint64)          # many equal depths: ties are broken by the low word
    keys = (depth << np.uint64(32)) | rng.permutation(n).astype(np.uint64)
    got, lst = sort_tile_model(keys)
    want = np.sort(keys)
    assert np.array_equal(got, want)
    assert np.array_equal(lst, (want & np.uint64(0xFFFFFFFF)).astype(np.uint32))


def test_chunk_table_matches_the_cap_clamped_lengths():
    """k_tile_scan: chunk counts are taken from the list lengths CLAMPED to the binning capacity, exactly as the
    sort kernels see them (min(tile_start, cap)); tiles of <= SORT_SHORT keys have no chunks."""
    counts = np.array([0, 100, 2048, 2049, 4096, 4097, 9000, 3000, 70000], np.int64)
    start = np.concatenate([[0], np.cumsum(counts)])
    for cap in (int(start[-1]), 15000, 5000):
        clamp = np.minimum(start, cap)
        nclamp = clamp[1:] - clamp[:-1]
        nch = np.where(nclamp > SORT_SHORT, (nclamp + SORT_CHUNK - 1) // SORT_CHUNK, 0)
        chunk_start = np.concatenate([[0], np.cumsum(nch)])
        # every chunk id maps back to (tile, k) with k < its tile's chunk count, by "last t with chunk_start[t] <= b"
        for b in range(int(chunk_start[-1])):
            t = int(np.searchsorted(chunk_start, b, side="right") - 1)
            assert nch[t] > 0 and 0 <= b - chunk_start[t] < nch[t]
        # upper bound used for the launch grid: cap / CHUNK + cap / SHORT + 2
        assert chunk_start[-1] <= cap // SORT_CHUNK + cap // SORT_SHORT + 2
