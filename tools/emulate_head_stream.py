#!/usr/bin/env python3
"""Lane-level numpy model of k_dpt_head_tail_s (csrc/ds_encoder_ops.hip): the index arithmetic of the streaming head-tail
kernel -- ring slots, the producers' staging tiles and lane -> (row, chunk) map, the consumers' fragment addresses, the MFMA operand
and accumulator layouts, the exchange of partial accumulators between the two channel halves, the segment / strip walk --
restated one to one and checked against the torch definition on the CPU.  It exists because the build container has no GPU:
the kernel's bookkeeping can be debugged here, and only its transcription to HIP remains to be checked on hardware
(tests/test_gpu_models.py::test_dpt_head_tail_kernel).  LDS starts as NaN, so a read of a slot nobody wrote shows up; every step
is run in both role orders (consumers first / producers first), which must agree because the roles touch disjoint ring rows.

    python tools/emulate_head_stream.py
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

TW, PW, RING, PIXB = 32, 34, 10, 272
NIT, SRC_ROWS, SRC_COLS = 9, 4, 7
CB, NCOLS = (0, 9, 18, 26), (9, 9, 8, 8)
f32 = np.float32


def mfma_32x32x16(areg, breg, acc):
    """areg, breg: [64 lanes, 8] (lane = 32 hi + l31); acc: [64, 16].  D[m][n] += sum_k A[m][k] B[k][n] with
    A[m][8 h + j] = areg[32 h + m][j], B[8 h + j][n] = breg[32 h + n][j], acc[32 h + n][r] = D[(r & 3) + 8 (r >> 2) + 4 h][n]."""
    A = np.zeros((32, 16), np.float64)
    B = np.zeros((16, 32), np.float64)
    for h in range(2):
        A[:, 8 * h:8 * h + 8] = areg[32 * h:32 * h + 32].astype(np.float64)
        B[8 * h:8 * h + 8, :] = breg[32 * h:32 * h + 32].astype(np.float64).T
    D = A @ B
    for h in range(2):
        for r in range(16):
            acc[32 * h:32 * h + 32, r] += D[(r & 3) + 8 * (r >> 2) + 4 * h, :].astype(f32)


class Model:
    def __init__(self, x, wf, b2, w3, b3, relu_out, oh, ow, ncu=3, seg_rows=None, producers_first=False):
        self.x = x                      # [B, ih, iw, 128] float16
        self.B, self.ih, self.iw, _ = x.shape
        self.wf = wf                    # [9, 8, 2, 32, 8] float16
        self.b2, self.w3, self.b3, self.relu_out = b2, w3, f32(b3), relu_out
        self.oh, self.ow = oh, ow
        self.sy = f32((self.ih - 1) / (oh - 1)) if oh > 1 else f32(0)
        self.sx = f32((self.iw - 1) / (ow - 1)) if ow > 1 else f32(0)
        self.strips_x = (ow + TW - 1) // TW
        sr = (oh + 3) & ~3
        while self.strips_x * self.B * ((oh + sr - 1) // sr) < 4 * ncu and sr > 16:
            sr = ((sr // 2) + 3) & ~3
        if seg_rows:
            sr = (seg_rows + 3) & ~3
        self.seg_rows = sr
        self.nseg = (oh + sr - 1) // sr
        self.nitems = self.strips_x * self.nseg * self.B
        self.grid = min(ncu, self.nitems)
        self.out = np.full((self.B, oh, ow), np.nan, f32)
        self.producers_first = producers_first

    # ---- producers -------------------------------------------------------------------------------------------------------
    def ylo(self, y0seg, jfirst):
        return min(int(f32(self.sy * f32(max(y0seg + jfirst, 0)))), self.ih - 1)

    def cols(self, tx0, w4):
        cb, ncols = CB[w4], NCOLS[w4]
        X = dict(cx0=[0] * NIT, cx1=[0] * NIT, wx0=[f32(0)] * NIT, wx1=[f32(0)] * NIT)
        X["xlo"] = min(int(f32(self.sx * f32(max(tx0 - 1 + cb, 0)))), self.iw - 1)
        for k in range(NIT):
            ox = tx0 - 1 + cb + k
            inside = k < ncols and 0 <= ox < self.ow
            fx = f32(self.sx * f32(max(ox, 0)))
            x0 = min(int(fx), self.iw - 1)
            x1 = min(x0 + 1, self.iw - 1)
            tx = f32(fx - f32(x0))
            X["cx0"][k], X["cx1"][k] = x0 - X["xlo"], x1 - X["xlo"]
            X["wx0"][k] = f32(1) - tx if inside else f32(0)
            X["wx1"][k] = tx if inside else f32(0)
            if k < ncols:
                assert 0 <= X["cx0"][k] < SRC_COLS and 0 <= X["cx1"][k] < SRC_COLS, "staging tile too narrow"
        return X

    def src_load(self, b, ylo, xlo):
        """registers of one producer wave: [7 loads][64 lanes][8 halves]"""
        R = np.zeros((SRC_ROWS * SRC_COLS * 16 // 64, 64, 8), np.float16)
        for m in range(R.shape[0]):
            for lane in range(64):
                idx = lane + 64 * m
                chunk, pixel = idx & 15, idx >> 4
                r, cx = divmod(pixel, SRC_COLS)
                y, x = min(ylo + r, self.ih - 1), min(xlo + cx, self.iw - 1)
                R[m, lane] = self.x[b, y, x, chunk * 8:chunk * 8 + 8]
        return R

    def produce(self, b, w4, X, y0seg, jfirst, nrows, R):
        cb, ncols = CB[w4], NCOLS[w4]
        stage = self.s_stage[w4]                                    # [28 pixels * 16 chunks][8 halves]
        for m in range(R.shape[0]):
            for lane in range(64):
                stage[lane + 64 * m] = R[m, lane]
        ylo = self.ylo(y0seg, jfirst)
        for lane in range(64):
            chunk, q = lane & 15, lane >> 4
            if q >= nrows:
                continue
            oy = y0seg + jfirst + q
            inside_y = 0 <= oy < self.oh
            fy = f32(self.sy * f32(max(oy, 0)))
            y0 = min(int(fy), self.ih - 1)
            y1 = min(y0 + 1, self.ih - 1)
            ty = f32(fy - f32(y0))
            wy0 = f32(1) - ty if inside_y else f32(0)
            wy1 = ty if inside_y else f32(0)
            assert 0 <= y0 - ylo < SRC_ROWS and 0 <= y1 - ylo < SRC_ROWS, "staging tile too short"
            slot = (jfirst + 1 + q) % RING
            for k in range(ncols):
                def rd(ry, cx):
                    return stage[(ry * SRC_COLS + cx) * 16 + chunk].astype(np.float64)
                a, bq = rd(y0 - ylo, X["cx0"][k]), rd(y0 - ylo, X["cx1"][k])
                cq, d = rd(y1 - ylo, X["cx0"][k]), rd(y1 - ylo, X["cx1"][k])
                o = f32(wy0 * X["wx0"][k]) * a + f32(wy0 * X["wx1"][k]) * bq + f32(wy1 * X["wx0"][k]) * cq + f32(wy1 * X["wx1"][k]) * d
                addr = (slot * PW + cb + k) * PIXB + chunk * 16
                self.s_act[addr // 2:addr // 2 + 8] = o.astype(f32).astype(np.float16)

    # ---- consumers -------------------------------------------------------------------------------------------------------
    def lanes(self):
        lane = np.arange(64)
        return lane >> 5, lane & 31

    def step(self, wave, t, acc0, acc1):
        kh, rp = wave & 1, wave >> 1
        hi, l31 = self.lanes()
        acc0[:] = 0
        acc1[:] = 0
        own = acc1 if kh else acc0                                   # the row this wave finishes starts at the 3x3 bias
        for r in range(16):
            own[:, r] = self.b2[(r & 3) + 8 * (r >> 2) + 4 * hi]
        sb = (4 * t + 2 * rp) % RING
        rowp = []
        for i in range(4):
            sl = sb + i
            sl = sl - RING if sl >= RING else sl
            rowp.append((sl * PW + l31) * PIXB + (8 * kh + hi) * 16)
        for g in range(12):
            dx, s4 = g >> 2, g & 3
            off = dx * PIXB + s4 * 32
            bq = []
            for i in range(4):
                ad = (rowp[i] + off) // 2
                bq.append(np.stack([self.s_act[a:a + 8] for a in ad]))
            for acc, dy, i in ((acc0, 0, 0), (acc1, 0, 1), (acc0, 1, 1), (acc1, 1, 2), (acc0, 2, 2), (acc1, 2, 3)):
                areg = self.wf[dy * 3 + dx, 4 * kh + s4][hi, l31]          # [64, 8]
                mfma_32x32x16(areg, bq[i], acc)

    def give(self, wave, t, acc0, acc1):
        kh = wave & 1
        g = acc0 if kh else acc1
        self.s_part[t & 1, wave] = g.reshape(64, 4, 4).transpose(1, 0, 2)   # [quad][lane][4]

    def finish(self, b, wave, t_prev, acc0, acc1, y0seg, yend, tx0):
        kh, rp = wave & 1, wave >> 1
        hi, l31 = self.lanes()
        fin = acc1 if kh else acc0
        part = np.zeros(64, f32)
        pp = self.s_part[t_prev & 1, wave ^ 1]                             # [quad][lane][4]
        for q in range(4):
            for j in range(4):
                ch = 8 * q + 4 * hi + j
                v = np.maximum(fin[:, 4 * q + j] + pp[q, :, j], f32(0))
                part = (part + self.w3[ch] * v).astype(f32)
        part = part + part[np.arange(64) ^ 32]
        res = part + self.b3
        if self.relu_out:
            res = np.maximum(res, f32(0))
        oy = y0seg + 4 * t_prev + 2 * rp + kh
        ox = tx0 + l31
        for ln in range(64):
            if hi[ln] == 0 and oy < yend and ox[ln] < self.ow:
                assert np.isnan(self.out[b, oy, ox[ln]]), "pixel written twice"
                self.out[b, oy, ox[ln]] = res[ln]

    # ---- the workgroup --------------------------------------------------------------------------------------------------
    def run(self):
        per_img = self.strips_x * self.nseg
        for wg in range(self.grid):
            self.s_act = np.full(RING * PW * PIXB // 2, np.nan, np.float16)
            self.s_part = np.full((2, 4, 4, 64, 4), np.nan, f32)
            self.s_stage = np.full((4, SRC_ROWS * SRC_COLS * 16, 8), np.nan, np.float16)
            acc = [(np.zeros((64, 16), f32), np.zeros((64, 16), f32)) for _ in range(4)]
            for item in range(wg, self.nitems, self.grid):
                b, rem = divmod(item, per_img)
                sxi, seg = divmod(rem, self.nseg)
                y0seg, tx0 = seg * self.seg_rows, sxi * TW
                yend = min(y0seg + self.seg_rows, self.oh)
                nsteps = (yend - y0seg + 3) >> 2
                X = [self.cols(tx0, w4) for w4 in range(4)]
                R = [self.src_load(b, self.ylo(y0seg, -1), X[w4]["xlo"]) for w4 in range(4)]
                for w4 in range(4):                                  # rows -1, 0, then rows 1 .. 4; the next tile is loaded in between
                    self.produce(b, w4, X[w4], y0seg, -1, 2, R[w4])
                    R[w4] = self.src_load(b, self.ylo(y0seg, 1), X[w4]["xlo"])
                    self.produce(b, w4, X[w4], y0seg, 1, 4, R[w4])
                    R[w4] = self.src_load(b, self.ylo(y0seg, 5), X[w4]["xlo"])
                for t in range(nsteps + 1):
                    def consumers():
                        if t > 0:
                            for w in range(4):
                                self.finish(b, w, t - 1, acc[w][0], acc[w][1], y0seg, yend, tx0)
                        if t < nsteps:
                            for w in range(4):
                                self.step(w, t, acc[w][0], acc[w][1])
                            for w in range(4):
                                self.give(w, t, acc[w][0], acc[w][1])

                    def producers():
                        if t + 1 < nsteps:
                            for w4 in range(4):
                                self.produce(b, w4, X[w4], y0seg, 4 * t + 5, 4, R[w4])
                                R[w4] = self.src_load(b, self.ylo(y0seg, 4 * t + 9), X[w4]["xlo"])
                    if self.producers_first:
                        producers(); consumers()
                    else:
                        consumers(); producers()
        return self.out


def reference(x, conv3, conv1, oh, ow, relu):
    with torch.no_grad():
        up = F.interpolate(x.float(), size=(oh, ow), mode="bilinear", align_corners=True)
        y = conv1(F.relu(conv3(up)))
        return (F.relu(y) if relu else y)[:, 0].numpy()


def main():
    torch.manual_seed(4)
    conv3 = torch.nn.Conv2d(128, 32, 3, padding=1)
    conv1 = torch.nn.Conv2d(32, 1, 1)
    with torch.no_grad():
        conv1.bias.fill_(0.05)
    worst = 0.0
    cases = [(2, 9, 13, 18, 26, True, None), (1, 20, 31, 33, 70, True, None), (1, 12, 12, 40, 37, False, 16), (3, 16, 16, 32, 32, False, 8)]
    for (b, ih, iw, oh, ow, relu, seg) in cases:
        x = torch.randn((b, 128, ih, iw)).half()
        want = reference(x, conv3, conv1, oh, ow, relu)
        w = conv3.weight.detach().half()
        wf = w.permute(2, 3, 1, 0).reshape(9, 8, 2, 8, 32).permute(0, 1, 2, 4, 3).contiguous().numpy()
        xn = x.permute(0, 2, 3, 1).contiguous().numpy()
        outs = []
        for pf in (False, True):
            m = Model(xn, wf, conv3.bias.detach().numpy(), conv1.weight.detach().reshape(32).numpy(), float(conv1.bias.item()), relu, oh, ow,
                      ncu=3, seg_rows=seg, producers_first=pf)
            outs.append(m.run())
        assert not np.isnan(outs[0]).any(), "pixels never written"
        assert np.array_equal(outs[0], outs[1]), "role order changes the result: producers and consumers share a ring row"
        err = float(np.abs(outs[0] - want).max())
        tol = 4e-3 * (1 + float(np.abs(want).max()))
        print(f"case {(b, ih, iw, oh, ow, relu, seg)}: items {m.nitems} seg_rows {m.seg_rows} max|err| {err:.2e} (tol {tol:.2e})")
        assert err < tol
        worst = max(worst, err)
    print("OK", worst)


if __name__ == "__main__":
    sys.exit(main())
