"""Lane-level numpy model of csrc/c3_fused32.hip (index logic only, float64, no rounding): every LDS address, fragment
layout, lane swap and store address of the kernel is restated per lane and the result is compared with a direct evaluation of
cv3(cat(x1 + m.cv2(m.cv1(x1)), cv2(x))).  Runs on the CPU; it checks the BOOK-KEEPING of the kernel (which lane holds which
pixel / channel, which LDS byte a fragment comes from), not the arithmetic of the MFMA -- the layouts assumed for
v_mfma_f32_32x32x16 are the ones every other conv kernel of this repo is tested with on the GPU:
  A fragment (weights):      lane (hi, r) holds row r,    k = 8*hi .. 8*hi+7 of the k16 step
  B fragment (activations):  lane (hi, r) holds pixel r,  k = 8*hi .. 8*hi+7
  D (swapped form):          acc[g*4 + e] of lane (hi, r) = cout g*8 + hi*4 + e of pixel r

    python tools/c3_fused_index_model.py            # prints the max abs difference for a few ragged shapes
"""
import numpy as np

TH = TW = 16
PW, PPIX, HG, SLOT, ROW = 18, 324, 11, 80, 1536
F3_W12, F3_WM1, F3_WM2, F3_W3 = 0, 8 * 1024, 10 * 1024, 28 * 1024
F3_BIAS = 36 * 1024
F3_PATCH = F3_BIAS + 6 * 128
LANES = np.arange(64)
HI, FROW = LANES >> 5, LANES & 31


def silu(v):
    return v / (1.0 + np.exp(-v))


class Lds:
    """byte-addressed LDS holding float64 'halfs' (2 bytes each) and fp32 words (4 bytes each) in separate planes"""

    def __init__(self, nbytes):
        self.h = np.full(nbytes // 2, np.nan)
        self.f = np.full(nbytes // 4, np.nan)

    def write16(self, byte, vals8):   # a 16-byte packet = 8 halfs
        assert byte % 16 == 0
        self.h[byte // 2: byte // 2 + 8] = vals8

    def read16(self, byte):
        assert byte % 16 == 0
        v = self.h[byte // 2: byte // 2 + 8]
        assert not np.isnan(v).any(), f"LDS read of unwritten bytes at {byte}"
        return v.copy()


def mfma(a_frag, b_frag, acc):
    """a_frag, b_frag: [64 lanes][8]; acc: [64][16] -> D[cout][pix] += A[cout][k] B[k][pix]"""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for l in range(64):
        A[FROW[l], 8 * HI[l]: 8 * HI[l] + 8] = a_frag[l]
        B[8 * HI[l]: 8 * HI[l] + 8, FROW[l]] = b_frag[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        for g in range(4):
            for e in range(4):
                out[l, g * 4 + e] += D[g * 8 + HI[l] * 4 + e, FROW[l]]
    return out


def permlane32_swap(a, b):
    """v_permlane32_swap: the upper half of a is exchanged with the lower half of b; a, b: [64][...]"""
    a2, b2 = a.copy(), b.copy()
    a2[32:] = b[:32]
    b2[:32] = a[32:]
    return a2, b2


def silu_pack_subtile(acc, rv, res):
    """conv_common.hpp: o[gq] of lane (hi, r) = words {rx0, ry0, rx1, ry1}; a word = 2 channels"""
    o = np.zeros((2, 64, 4, 2))
    for g in (0, 2):
        pk = np.zeros((2, 2, 64, 2))
        for h in range(2):
            for p in range(2):
                v = np.stack([acc[:, (g + h) * 4 + 2 * p], acc[:, (g + h) * 4 + 2 * p + 1]], axis=1)
                v = silu(v)
                if res:
                    v = v + rv[g + h][:, p, :]
                pk[h][p] = v
        rx0, rx1 = permlane32_swap(pk[0][0], pk[1][0])
        ry0, ry1 = permlane32_swap(pk[0][1], pk[1][1])
        o[g >> 1] = np.stack([rx0, ry0, rx1, ry1], axis=1)
    return o   # [gq][lane][word][2]


def unswap_packets(o):
    rv = np.zeros((4, 64, 2, 2))
    for gq in range(2):
        rx0, rx1 = permlane32_swap(o[gq][:, 0], o[gq][:, 2])
        ry0, ry1 = permlane32_swap(o[gq][:, 1], o[gq][:, 3])
        rv[2 * gq] = np.stack([rx0, ry0], axis=1)
        rv[2 * gq + 1] = np.stack([rx1, ry1], axis=1)
    return rv   # [g][lane][p][2]


def as_frag(packet):   # [lane][4 words][2] -> [lane][8]
    return packet.reshape(64, 8)


def run_kernel(x, w12, b12, wm1, bm1, wm2, bm2, w3, b3):
    """x: (n, h, w, 64); packed weights [rows][k] (K-major, k = (ky*3 + kx)*32 + c for the 3x3); returns y (n, h, w, 64)"""
    n, h, w, _ = x.shape
    y = np.full((n, h, w, 64), np.nan)
    lds = Lds(F3_PATCH + (TH + 2) * ROW)
    # ---- resident weights / biases ----
    def fill(base, wt, nfrag, ks):
        for f in range(nfrag):
            t, s = divmod(f, ks)
            for l in range(64):
                lds.write16(base + (f * 64 + l) * 16, wt[t * 32 + FROW[l], 16 * s + 8 * HI[l]: 16 * s + 8 * HI[l] + 8])
    fill(F3_W12, w12, 8, 4)
    fill(F3_WM1, wm1, 2, 2)
    fill(F3_WM2, wm2, 18, 18)
    fill(F3_W3, w3, 8, 4)
    bl = np.zeros((48, 4))
    for tid in range(48):
        t, g, hh = tid >> 3, (tid >> 1) & 3, tid & 1
        src = b12[t * 32:] if t < 2 else (bm1 if t == 2 else (bm2 if t == 3 else b3[(t - 4) * 32:]))
        bl[tid] = src[g * 8 + hh * 4: g * 8 + hh * 4 + 4]

    def wfrag(base, f):
        return np.stack([lds.read16(base + (f * 64 + l) * 16) for l in range(64)])

    def bias_acc(tile):
        acc = np.zeros((64, 16))
        for l in range(64):
            for g in range(4):
                acc[l, g * 4: g * 4 + 4] = bl[(tile * 4 + g) * 2 + HI[l]]
        return acc

    tiles_x, tiles_y = -(-w // TW), -(-h // TH)
    for img in range(n):
        for ty in range(tiles_y):
            for tx in range(tiles_x):
                oy0, ox0 = ty * TH, tx * TW
                lds.h[F3_PATCH // 2:] = np.nan   # (model only: catches reads of slots this tile did not write)
                pend = []   # patch writes of all 8 waves (between the two barriers)
                waves = []
                for wave in range(8):
                    pr_o, pc_o = (wave * 32 + FROW) // TW, (wave * 32 + FROW) % TW
                    oy, ox = oy0 + pr_o, ox0 + pc_o
                    okc = (oy < h) & (ox < w)
                    cy, cx = np.minimum(oy, h - 1), np.minimum(ox, w - 1)
                    xb = [np.stack([x[img, cy[l], cx[l], 8 * HI[l] + 16 * s: 8 * HI[l] + 16 * s + 8] for l in range(64)]) for s in range(4)]

                    def halo(hg):
                        q = hg * 32 + FROW
                        qc = np.minimum(q, PPIX - 1)
                        pr, pc = qc // PW, qc % PW
                        iy, ix = oy0 - 1 + pr, ox0 - 1 + pc
                        inside = (q < PPIX) & (iy >= 0) & (iy < h) & (ix >= 0) & (ix < w)
                        cy2, cx2 = np.clip(iy, 0, h - 1), np.clip(ix, 0, w - 1)
                        xa = [np.stack([x[img, cy2[l], cx2[l], 8 * HI[l] + 16 * s: 8 * HI[l] + 16 * s + 8] for l in range(64)]) for s in range(4)]
                        acc = bias_acc(0)
                        for s in range(4):
                            acc = mfma(wfrag(F3_W12, s), xa[s], acc)
                        p1 = silu_pack_subtile(acc, None, False)
                        acc2 = bias_acc(2)
                        for s in range(2):
                            acc2 = mfma(wfrag(F3_WM1, s), as_frag(p1[s]), acc2)
                        pu = silu_pack_subtile(acc2, None, False)
                        pu[:, ~inside] = 0.0
                        po = np.where(q < PPIX, pr * ROW + pc * SLOT + HI * 16, -1)
                        return po, pu

                    q0, pu0 = halo(wave)
                    for gq in range(2):
                        for l in range(64):
                            pend.append((F3_PATCH + q0[l] + gq * 32, as_frag(pu0[gq])[l]))
                    if wave + 8 < HG:
                        q1, pu1 = halo(wave + 8)
                        for gq in range(2):
                            for l in range(64):
                                if q1[l] >= 0:
                                    pend.append((F3_PATCH + q1[l] + gq * 32, as_frag(pu1[gq])[l]))
                    acc0, acc1 = bias_acc(0), bias_acc(1)
                    for s in range(4):
                        acc0 = mfma(wfrag(F3_W12, s), xb[s], acc0)
                        acc1 = mfma(wfrag(F3_W12, 4 + s), xb[s], acc1)
                    pk1, pk2 = silu_pack_subtile(acc0, None, False), silu_pack_subtile(acc1, None, False)
                    waves.append((pr_o, pc_o, oy, ox, okc, pk1, pk2))
                for byte, v in pend:
                    lds.write16(byte, v)
                for wave in range(8):
                    pr_o, pc_o, oy, ox, okc, pk1, pk2 = waves[wave]
                    pc_base = F3_PATCH + pr_o * ROW + pc_o * SLOT + HI * 16
                    acc = bias_acc(3)
                    for ts in range(18):
                        off = ((ts >> 1) // 3) * ROW + ((ts >> 1) % 3) * SLOT + (ts & 1) * 32
                        fa = np.stack([lds.read16(pc_base[l] + off) for l in range(64)])
                        acc = mfma(wfrag(F3_WM2, ts), fa, acc)
                    rv = unswap_packets(pk1)
                    pv = silu_pack_subtile(acc, rv, True)
                    a0, a1 = bias_acc(4), bias_acc(5)
                    for s in range(4):
                        xf = as_frag(pv[s] if s < 2 else pk2[s - 2])
                        a0 = mfma(wfrag(F3_W3, s), xf, a0)
                        a1 = mfma(wfrag(F3_W3, 4 + s), xf, a1)
                    o0, o1 = silu_pack_subtile(a0, None, False), silu_pack_subtile(a1, None, False)
                    for l in range(64):
                        if okc[l]:
                            for q in range(2):
                                c = HI[l] * 8 + q * 16
                                y[img, oy[l], ox[l], c: c + 8] = as_frag(o0[q])[l]
                                y[img, oy[l], ox[l], 32 + c: 32 + c + 8] = as_frag(o1[q])[l]
    return y


def reference(x, w12, b12, wm1, bm1, wm2, bm2, w3, b3):
    n, h, w, _ = x.shape
    t12 = silu(x @ w12[:64].T + b12)               # (n, h, w, 64): x1 | x2
    x1, x2 = t12[..., :32], t12[..., 32:]
    u = silu(x1 @ wm1[:32].T + bm1)
    up = np.zeros((n, h + 2, w + 2, 32))
    up[:, 1:-1, 1:-1] = u
    acc = np.zeros((n, h, w, 32)) + bm2
    for ky in range(3):
        for kx in range(3):
            acc += up[:, ky: ky + h, kx: kx + w] @ wm2[:32, (ky * 3 + kx) * 32: (ky * 3 + kx) * 32 + 32].T
    v = x1 + silu(acc)
    return silu(np.concatenate([v, x2], axis=-1) @ w3[:64].T + b3)


def main():
    rng = np.random.default_rng(0)
    for n, h, w in [(1, 16, 16), (2, 21, 37), (1, 5, 3), (1, 33, 16)]:
        x = rng.standard_normal((n, h, w, 64))
        w12, wm1 = rng.standard_normal((128, 64)) / 8, rng.standard_normal((128, 32)) / 6
        wm2, w3 = rng.standard_normal((128, 288)) / 17, rng.standard_normal((128, 64)) / 8
        b12, bm1, bm2, b3 = rng.standard_normal(64), rng.standard_normal(32), rng.standard_normal(32), rng.standard_normal(64)
        got = run_kernel(x, w12, b12, wm1, bm1, wm2, bm2, w3, b3)
        ref = reference(x, w12, b12, wm1, bm1, wm2, bm2, w3, b3)
        assert not np.isnan(got).any(), "an output pixel was never stored"
        print(f"n={n} {h}x{w}: max |model - reference| = {np.abs(got - ref).max():.3e}")
        assert np.abs(got - ref).max() < 1e-9


if __name__ == "__main__":
    main()
