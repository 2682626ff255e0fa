#!/usr/bin/env python3
"""Tile-level numpy model of the control flow of k_attention_fwd3 (csrc/ds_attention.hip, -DDS_EXPERIMENTS builds): the attention
kernel generation that computes S of tile t + 1 beside the softmax of tile t inside one wave.  What it restates one to one:
the prologue, which K / V^T tile sits in which LDS buffer at which time (K runs one tile ahead of V^T), the two alternating S
accumulator sets, the peeled last iteration with its key mask, the deferred running maximum, the order rescale -> P.V.  What it
checks: every LDS read sees the tile the algorithm means (buffers start as NaN and carry a tag), no buffer is written in the
barrier interval in which it is read (the cross-wave hazard the one-barrier-per-tile schedule must not have), and the output
equals softmax attention.  The build container has no GPU: the kernel's bookkeeping is debugged here, its transcription to HIP
on hardware (tools/att_ab.sh: bit-identical to generation 2 at the benchmark shape, profiles/round4_attention_gen3_ab.txt).

    python tools/emulate_attention_skew.py
"""
import numpy as np

KB = 64
THR = 6.0          # AT2_THR, log2 units


class Lds:
    """K[2] | V^T[2]; every buffer remembers which tile it holds and in which barrier interval it was last read / written"""

    def __init__(self):
        self.data = {("K", 0): None, ("K", 1): None, ("V", 0): None, ("V", 1): None}
        self.tag = {k: None for k in self.data}
        self.read_at = {k: -10 for k in self.data}
        self.written_at = {k: -10 for k in self.data}
        self.interval = 0

    def barrier(self):
        self.interval += 1

    def stash(self, kind, buf, tile, value):
        key = (kind, buf)
        assert self.read_at[key] != self.interval, f"{key} is written in the interval in which it is read (tile {tile})"
        self.data[key], self.tag[key] = value, tile
        self.written_at[key] = self.interval

    def read(self, kind, buf, tile):
        key = (kind, buf)
        assert self.tag[key] == tile, f"{key} holds tile {self.tag[key]}, the kernel means tile {tile}"
        assert self.written_at[key] != self.interval, f"{key} is read in the interval in which it is written (tile {tile})"
        self.read_at[key] = self.interval
        return self.data[key]


def attention_wave(q, k, v, bias, n_valid, scale):
    """q [32, 64] (one wave's queries), k / v [Np64, 64], bias [32, Np64] or None (natural units) -> out [32, 64], float32 math
    with the probabilities rounded to float16 before P.V as in the kernel"""
    f32 = np.float32
    ntiles = (n_valid + KB - 1) // KB
    c = f32(scale * 1.4426950408889634)
    thr_x = f32(THR) / c
    lds = Lds()

    def k_tile(t):
        return k[t * KB:(t + 1) * KB].astype(f32)

    def v_tile(t):
        return v[t * KB:(t + 1) * KB].astype(f32)

    def s_tile(t):                                  # A3_S: bias MFMAs start the chain (bias / scale through the identity), then K . Q^T
        kt = lds.read("K", t & 1, t)
        s = np.zeros((32, KB), f32)
        if bias is not None:
            s += (bias[:, t * KB:(t + 1) * KB] / scale).astype(np.float16).astype(f32)
        return s + q.astype(f32) @ kt.T

    o = np.zeros((32, 64), f32)
    m_run = np.full(32, -np.inf, f32)
    l_run = np.zeros(32, f32)
    # prologue: K(0), V^T(0) [, K(1)] into LDS | barrier | S(0)
    lds.stash("K", 0, 0, k_tile(0))
    lds.stash("V", 0, 0, v_tile(0))
    if ntiles > 1:
        lds.stash("K", 1, 1, k_tile(1))
    lds.barrier()
    acc = {"A": s_tile(0), "B": None}
    lds.barrier()                                   # iteration 0 overwrites K(0) with K(2): everybody is done with S(0) first

    def iteration(cur, nxt, t, has_next, masked):
        nonlocal o, m_run, l_run
        more2 = t + 2 < ntiles
        fetched_k = k_tile(t + 2) if more2 else None          # A3_FETCH_K(t + 2): registers, in flight during the iteration
        fetched_v = v_tile(t + 1) if has_next else None       # A3_FETCH_V(t + 1)
        s = acc[cur]
        if masked:
            keys = t * KB + np.arange(KB)
            s = np.where(keys[None, :] >= n_valid, -np.inf, s).astype(f32)
        if has_next:
            acc[nxt] = s_tile(t + 1)                          # phase A: beside the maxima and the first half of the exponentials
        mx = s.max(axis=1)
        grow = mx > m_run + thr_x
        mn = np.where(grow, mx, m_run).astype(f32)
        alpha = np.exp2((m_run - mn) * c).astype(f32)
        m_run = mn
        p = np.exp2(s * c - (mn * c)[:, None]).astype(f32)
        if grow.any():
            o = o * alpha[:, None]
        vt = lds.read("V", t & 1, t)
        o = o + p.astype(np.float16).astype(f32) @ vt         # phases B and C
        l_run = l_run * alpha + p.sum(axis=1)
        if more2:
            lds.stash("K", t & 1, t + 2, fetched_k)
        if has_next:
            lds.stash("V", (t + 1) & 1, t + 1, fetched_v)
        lds.barrier()

    pad_keys = (n_valid & (KB - 1)) != 0
    ntl = ntiles - 1
    t = 0
    while t + 1 < ntl:
        iteration("A", "B", t, True, False)
        iteration("B", "A", t + 1, True, False)
        t += 2
    if t < ntl:
        iteration("A", "B", t, True, False)
        iteration("B", "A", ntl, False, pad_keys)
    else:
        iteration("A", "B", ntl, False, pad_keys)
    return o / l_run[:, None]


def reference(q, k, v, bias, n_valid, scale):
    s = q.astype(np.float64) @ k[:n_valid].astype(np.float64).T * scale
    if bias is not None:
        s = s + bias[:, :n_valid].astype(np.float64)
    s -= s.max(axis=1, keepdims=True)
    p = np.exp(s)
    return (p / p.sum(axis=1, keepdims=True)) @ v[:n_valid].astype(np.float64)


def check(n_valid, with_bias, seed=0):
    rng = np.random.default_rng(seed)
    np64 = (n_valid + 63) // 64 * 64
    q = rng.standard_normal((32, 64)).astype(np.float16)
    k = rng.standard_normal((np64, 64)).astype(np.float16)
    v = rng.standard_normal((np64, 64)).astype(np.float16)
    k[n_valid:] = 7.0                                           # junk in the pad rows: must not matter
    v[n_valid:] = -9.0
    bias = (rng.standard_normal((32, np64)) * 2).astype(np.float32) if with_bias else None
    got = attention_wave(q, k, v, bias, n_valid, 0.125)
    want = reference(q, k, v, bias, n_valid, 0.125)
    return float(np.abs(got - want).max() / (1 + np.abs(want).max()))


if __name__ == "__main__":
    for n in (1, 17, 64, 65, 128, 129, 191, 192, 193, 320, 1025):
        for wb in (False, True):
            print(f"n_valid {n:5d} bias {int(wb)}: max relative error {check(n, wb):.2e}")
