"""CPU model of csrc/ds_linear.hip (maintenance tool, runs without a GPU).

1. `check_schedule()`  -- the 8-phase K loop as barrier intervals: every fragment read must see a half-tile whose
   LDS-DMA has been waited for (counted vmcnt) by BOTH wave-rows before a barrier the reader has passed, and no
   half-tile may be re-staged before every read of its previous content was retired behind a barrier.
2. `check_indexing()` -- the address arithmetic (source swizzle of the DMA, LDS image, swizzled fragment reads, MFMA
   32x32x16 operand/accumulator layout, epilogue) executed with numpy for one workgroup; must equal x @ W.T.

    python tools/linear_model.py
"""
import numpy as np

# ---- 1. schedule ---------------------------------------------------------------------------------------------------
# phase j of iteration i (tiles E = 2i, O = 2i+1): (read kind, read tile offset, read buffer), (stage kind, tile, buffer)
READS = {0: ('B0', 0, 0), 1: ('B1', 0, 0), 2: ('A1', 0, 0), 3: ('A0', 1, 1),
         4: ('B0', 1, 1), 5: ('B1', 1, 1), 6: ('A1', 1, 1), 7: ('A0', 2, 0)}
STAGES = {0: ('A1', 1, 1), 1: ('A0', 2, 0), 2: ('B0', 2, 0), 3: ('B1', 2, 0),
          4: ('A1', 2, 0), 5: ('A0', 3, 1), 6: ('B0', 3, 1), 7: ('B1', 3, 1)}
LAST_WAIT = {0: 10, 1: 8, 2: 6, 3: 4, 4: 2, 5: 0}


def check_schedule(nt):
    ni = nt // 2
    # per wave-row g: list of (interval, op, ...) in program order
    issued = {0: [], 1: []}          # DMAs in issue order: dict(region, tile, issue_interval, land_interval)
    reads = []                       # dict(group, region, tile, issue_interval, retire_interval)

    def interval(g, phase, part):    # part 0 = memory part, 1 = MFMA part; barrier b[k] ends interval k
        return 2 * phase + part + g + 1  # +1: interval 0 is the prologue before the common barrier

    def wait(g, n, itv):
        lst = issued[g]
        for d in lst[:len(lst) - n if n else len(lst)]:
            if d['land'] is None:
                d['land'] = itv

    for g in (0, 1):
        for kind, t, s in (('A0', 0, 0), ('B0', 0, 0), ('B1', 0, 0), ('A1', 0, 0), ('A0', 1, 1), ('B0', 1, 1), ('B1', 1, 1)):
            issued[g].append(dict(region=(kind, s), tile=t, issue=0, land=None))
        wait(g, 5, 0)                # vmcnt(10) = 5 half-tiles of 2 loads stay in flight
        # prologue read of A0(tile 0) after the common barrier b[0]: interval 1 for both wave-rows (it precedes the
        # stagger barrier), retired by the lgkmcnt(0) that follows it at once
        reads.append(dict(group=g, region=('A0', 0), tile=0, issue=1, retire=1))
        for i in range(ni):
            last = i == ni - 1
            for j in range(8):
                p = 8 * i + j
                m_itv = interval(g, p, 0)
                if not (last and j == 7):
                    kind, dt, s = READS[j]
                    reads.append(dict(group=g, region=(kind, s), tile=2 * i + dt, issue=m_itv, retire=interval(g, p, 1)))
                if not last or j == 0:
                    kind, dt, s = STAGES[j]
                    issued[g].append(dict(region=(kind, s), tile=2 * i + dt, issue=m_itv, land=None))
                    wait(g, 5, m_itv)
                elif j in LAST_WAIT:
                    wait(g, LAST_WAIT[j] // 2, m_itv)
    problems = []
    for g in (0, 1):
        for d in issued[g]:
            assert d['tile'] < nt, ('stages a tile beyond K', d)
            if d['land'] is None:
                problems.append(('never waited for', g, d))
    for r in reads:
        # read-after-write: both wave-rows' DMA of (region, tile) landed in an interval strictly before the read's
        for g in (0, 1):
            ds = [d for d in issued[g] if d['region'] == r['region'] and d['tile'] == r['tile']]
            if len(ds) != 1:
                problems.append(('no unique staging', r, g))
                continue
            if ds[0]['land'] is None or not ds[0]['land'] < r['issue']:
                problems.append(('RAW', r, g, ds[0]))
        # write-after-read: any LATER staging of the region is issued strictly after the read was retired
        for g in (0, 1):
            for d in issued[g]:
                if d['region'] == r['region'] and d['tile'] > r['tile'] and not d['issue'] > r['retire']:
                    problems.append(('WAR', r, g, d))
                if d['region'] == r['region'] and d['tile'] < r['tile'] and not d['issue'] < r['issue']:
                    problems.append(('order', r, g, d))
    # every tile's 4 half-tiles are read exactly once by each wave-row
    for g in (0, 1):
        seen = sorted((r['tile'], r['region'][0]) for r in reads if r['group'] == g)
        want = sorted((t, k) for t in range(nt) for k in ('A0', 'A1', 'B0', 'B1'))
        if seen != want:
            problems.append(('coverage', g, set(want) ^ set(seen)))
    return problems


# ---- 2. indexing ---------------------------------------------------------------------------------------------------
LN_HALF, LN_B_BASE = 16384, 65536


def region_base(kind, s):
    k = {'A0': 0, 'A1': 1, 'B0': 2, 'B1': 3}[kind]
    return (k >> 1) * LN_B_BASE + (k & 1) * 2 * LN_HALF + s * LN_HALF


def stage(lds, kind, kt, s, x, w, bm0, bn0, M, K):
    """LDS-DMA of one half-tile: every thread (wave wid, lane) moves 2 x 16 bytes (8 halfs), lane-linear destination."""
    for wid in range(8):
        for i in range(2):
            c = 2 * wid + i
            for lane in range(64):
                j = 8 * c + (lane >> 3)
                slot = (lane & 7) ^ ((j >> 1) & 7)
                h = {'A0': 0, 'A1': 1, 'B0': 0, 'B1': 1}[kind]
                if kind[0] == 'A':
                    row = (j >> 6) * 128 + h * 64 + (j & 63)
                    src = x[bm0 + row, kt * 64 + slot * 8: kt * 64 + slot * 8 + 8]
                else:
                    col = (j >> 5) * 64 + (j & 31) + h * 32
                    src = w[bn0 + col, kt * 64 + slot * 8: kt * 64 + slot * 8 + 8]
                dst = region_base(kind, s) + c * 1024 + lane * 16
                lds[dst // 2: dst // 2 + 8] = src


def read_frag(lds, kind, s, wr, wc, rb, ks):
    """[64 lanes, 8] fragment of one ds_read_b128."""
    out = np.empty((64, 8), lds.dtype)
    h = int(kind[1])
    for lane in range(64):
        sl = ((2 * ks + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4
        if kind[0] == 'A':
            off = (wr * 64 + (lane & 31)) * 128 + sl + h * 2 * LN_HALF + s * LN_HALF + rb * 4096
        else:
            off = LN_B_BASE + (wc * 32 + (lane & 31)) * 128 + sl + h * 2 * LN_HALF + s * LN_HALF
        out[lane] = lds[off // 2: off // 2 + 8]
    return out


def mfma_32x32x16(a, b, c):
    """v_mfma_f32_32x32x16: a, b [64, 8]; lane l holds A[m = l&31][k = 8 (l>>5) + t], B[k][n = l&31];
    c [64, 16]: lane l, register r holds D[m = (r&3) + 8 (r>>2) + 4 (l>>5)][n = l&31]."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        for t in range(8):
            A[l & 31, 8 * (l >> 5) + t] = a[l, t]
            B[8 * (l >> 5) + t, l & 31] = b[l, t]
    D = A @ B
    out = c.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def check_indexing(M=300, N=256, K=128, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    y = np.zeros((M, N))
    nt = K // 64
    for bm0 in [min(p0, M - 256) for p0 in range(0, M, 256)]:      # the last row panel is shifted up to end at row M
        for bn0 in range(0, N, 256):
            lds = np.zeros(131072 // 2, np.float32)
            for wr in range(2):
                for wc in range(4):
                    acc = np.zeros((2, 2, 2, 64, 16))
                    for kt in range(nt):
                        s = kt & 1
                        for kind in ('A0', 'A1', 'B0', 'B1'):
                            stage(lds, kind, kt, s, x, w, bm0, bn0, M, K)
                        for ha in range(2):
                            for hb in range(2):
                                for ks in range(4):
                                    fb = read_frag(lds, 'B%d' % hb, s, wr, wc, 0, ks)
                                    for rb in range(2):
                                        fa = read_frag(lds, 'A%d' % ha, s, wr, wc, rb, ks)
                                        acc[ha, rb, hb] = mfma_32x32x16(fb, fa, acc[ha, rb, hb])
                    for hb in range(2):
                        for ha in range(2):
                            for rb in range(2):
                                blk = acc[ha, rb, hb]                       # [64 lanes, 16 registers]
                                for k in range(2):
                                    for lane in range(64):
                                        hi = lane >> 5
                                        # v_permlane32_swap(vdst = reg 8k+t, src = reg 8k+4+t): lanes 32-63 of vdst <-> lanes 0-31 of src
                                        v = np.empty(8)
                                        for t in range(4):
                                            a_own, b_own = blk[lane, 8 * k + t], blk[lane, 8 * k + 4 + t]
                                            if hi == 0:
                                                v[t], v[4 + t] = a_own, blk[lane + 32, 8 * k + t]
                                            else:
                                                v[t], v[4 + t] = blk[lane - 32, 8 * k + 4 + t], b_own
                                        m = bm0 + wr * 128 + ha * 64 + rb * 32 + (lane & 31)
                                        n0 = bn0 + wc * 64 + hb * 32 + 8 * hi + 16 * k
                                        if m < M:
                                            y[m, n0:n0 + 8] = v
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    return np.abs(y - ref).max()


def check_conv_indexing(B=1, H=11, W=25, C=128, N=256, seed=1):
    """The implicit 3x3 convolution of ds_conv3x3_nhwc: K-tile kt = tap kt % 9 of the 64-channel chunk kt / 9, the DMA source
    is the pixel shifted by the tap or the zero line (border bits of the thread's pixels), weights stay [out][tap][in]
    (K offset (tap * cpt + cc) * 64), the last row panel is shifted up.  One wave-level emulation per tile is enough here:
    the LDS image / fragment path is check_indexing's; this one pins the gather.  Returns max |error| vs a direct convolution."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, C))
    w = rng.standard_normal((N, 3, 3, C))
    M = B * H * W
    cpt = C // 64
    nt = 9 * cpt
    xf = x.reshape(M, C)
    wf = w.reshape(N, 9 * C)
    y = np.zeros((M, N))
    for bm0 in [min(p0, M - 256) for p0 in range(0, M, 256)]:
        acc = np.zeros((256, N))
        for kt in range(nt):
            cc = (kt * 7282) >> 16
            tap = kt - 9 * cc
            assert cc == kt // 9
            dy = ((tap * 11) >> 5) - 1
            dx = tap - 3 * (dy + 1) - 1
            a = np.zeros((256, 64))
            for row in range(256):
                pix = (bm0 + row) % (H * W)
                py, px = pix // W, pix % W
                bits = (1 if py > 0 else 0) | 2 | (4 if py < H - 1 else 0) | (8 if px > 0 else 0) | 16 | (32 if px < W - 1 else 0)
                ok = (bits >> (dy + 1)) & (bits >> (4 + dx)) & 1
                if ok:
                    src = bm0 + row + dy * W + dx                      # xb + srcA + aoff: the same pixel index, shifted
                    a[row] = xf[src, cc * 64: cc * 64 + 64]
            koff = (tap * cpt + cc) * 64
            acc += a @ wf[:, koff: koff + 64].T
        y[bm0: bm0 + 256] = acc
    ref = np.zeros((B, H, W, N))
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    for ky in range(3):
        for kx in range(3):
            ref += xp[:, ky: ky + H, kx: kx + W, :] @ w[:, ky, kx, :].T
    return np.abs(y - ref.reshape(M, N)).max()


def bank_conflicts():
    """16-lane groups of a fragment read must hit 16 distinct 16-byte slots of the 256-byte bank row."""
    worst = 0
    for ks in range(4):
        for grp in range(4):
            slots = set()
            for lane in range(16 * grp, 16 * grp + 16):
                off = (lane & 31) * 128 + (((2 * ks + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4)
                slots.add((off % 256) // 16)
            worst = max(worst, 16 - len(slots))
    return worst


if __name__ == '__main__':
    for nt in (2, 4, 6, 16, 64):
        p = check_schedule(nt)
        print('schedule nt=%d:' % nt, 'ok' if not p else p[:4])
    print('bank conflicts (missing slots per 16-lane group):', bank_conflicts())
    print('indexing max |err|:', check_indexing())
    print('convolution gather max |err|:', check_conv_indexing())
