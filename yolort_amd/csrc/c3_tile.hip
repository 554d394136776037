// A whole C3 block (or a slice of one: its Bottlenecks one at a time) per launch, for the 64- and 128-channel hidden widths of yolov5s'
// 80 x 80 and 40 x 40 levels (gfx950): y = cv3(cat(m(cv1(x)), cv2(x))), m = Bottleneck(s) x1 (+) cv2(cv1(x1)).
//
// Replaces yolort/v5/models/common.py:172-173 (C3.forward) with :115-116 (Bottleneck.forward) inlined, each Conv being
// common.py:69-70 (SiLU(BN(conv))) with the BatchNorm folded on the host -- for the hidden widths the resident-weights
// instance of c3_fused32.hip does not hold.
//
// Why: launched one convolution at a time the 40 x 40 / 80 x 80 half of the yolov5s stack runs at
// 0.2-0.4 of its per-layer bound: every launch is one block per CU whose prologue (cold operand fetch), pipeline ramp and
// write-back nothing overlaps, and every intermediate goes through memory.  Here
//   * a block of 8 waves owns a strip of R full-width output rows of one image; the pixels of the (R + 2)-row halo strip are laid
//     out as "patch slots" q = delta + r * (w + 1) + c -- ONE pad slot per row serves as the right padding of row r and the left
//     padding of row r + 1 -- so that a 3x3 tap is a CONSTANT slot shift, and a wave owns whole 32-slot groups (one MFMA pixel
//     column block) for ALL output channels of every convolution: nothing but the 3x3's input crosses waves;
//   * phase A (cv1 | cv2): x goes global -> VGPR (lane = pixel, 16 bytes per k16 step, like conv1x1_stream.hip); B (m.cv1): the
//     rounded output packets of A ARE its activation fragments (registers); its output t goes to an LDS patch [32-channel plane]
//     [slot][64 B] (zero outside the image: the 3x3's padding); C (m.cv2, 3x3): fragments from the patch at slot + shift, the
//     shortcut from A's packets; D (cv3): K = [C's packets | cv2's packets], both still in registers; 16-byte NHWC stores;
//   * ALL weights stream through ONE three-slot LDS ring as a single sequence of stages (A's k32 chunks, B's, the 3x3's
//     (chunk, kernel row) stages, D's k32 chunks, then the next tile's), pre-ordered on the host into MFMA fragment order
//     (ymi_c3_pack: a 1 KiB DMA piece is one 32-row x 16-k fragment, read back at lane * 16 + constant): the fetch of the next
//     phase / tile runs under the current one's MFMAs -- what separate launches cannot do.
// The halo costs (R + 2) / R of cv1 and m.cv1 (1/9 of the block's MACs).  Every intermediate is rounded to the storage dtype
// exactly where the separate launches round it and every accumulation runs in the same k order on top of the bias as the
// kernels it replaces (1x1: k ascending; 3x3: conv_halo8.hip's (chunk, dy, dx, k16) order), so the result is BIT-IDENTICAL
// to them (tests/test_hipsim_kernels.py on the CPU simulator, tests/test_c3_fused_gpu.py on the GPU).
//
// Modes (ymi_c3_desc.mode): 0 whole block (one Bottleneck); 1 HEAD = cv1 | cv2 + first Bottleneck (writes its output and
// cv2's); 2 MID = one Bottleneck; 3 TAIL = last Bottleneck + cv3 -- a C3 with n Bottlenecks is HEAD, (n - 2) x MID, TAIL.
#include "conv_common.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace ymi {

constexpr int C3T_MAXHJ = 6;

struct C3TArgs {
    const uint16_t* x;        // modes 0, 1: the block's input (n, h, w, cin)
    const uint16_t* y1_in;    // modes 2, 3: the Bottleneck's input (n, h, w, CH)
    const uint16_t* y2_in;    // mode 3: cv2(x) (n, h, w, CH)
    uint16_t* y;              // modes 0, 3: the block's output (n, h, w, 2 CH)
    uint16_t* y1_out;         // modes 1, 2: the Bottleneck's output
    uint16_t* y2_out;         // mode 1: cv2(x)
    const unsigned char* blob;   // ymi_c3_pack's stream: [A stages][B][C][D][bias fp32]
    int n, h, w, cin;
    int x_cs, y_cs, y1i_cs, y1o_cs, y2_cs;
    int mode, shortcut;
    int nst_a, nst_d;         // stages of phases A / D in this mode (0 when the phase is not part of it)
    int bias_off;             // byte offset of the bias section in the blob
};

struct C3TGeom {
    int R, pw, delta, nslot, tiles_per_img, ntiles;
    unsigned magic_pw;
    signed char grp[8][C3T_MAXHJ];    // per wave: the patch-slot group of each half-job (-1: none)
    signed char half[8][C3T_MAXHJ];   // 0: cv1's half of phase A (+ phase B); 1: cv2's half (its group is a centre group: phases C, D)
};

// CH = hidden width.  NHJ half-jobs per wave: (group, cv1 half) or (group, cv2 half) of phase A; centre job c is the pair (2c, 2c + 1).
template <int CH> struct C3TCfg;
template <> struct C3TCfg<128> {   // 40 x 40: a wave owns ONE centre group, or two halo-only groups
    static constexpr int NP = 4, NHJ = 2, GC = 1;
    static constexpr bool maybe_w1(int hj) { return true; }
    static constexpr int geom_of(int hj) { return hj; }           // (half-job 1 is either a second halo-only group or cv2's half of job 0's group: its own geometry)
};
template <> struct C3TCfg<64> {    // 80 x 80: a wave owns up to TWO centre groups and one halo-only group
    static constexpr int NP = 2, NHJ = 5, GC = 2;
    static constexpr bool maybe_w1(int hj) { return (hj & 1) == 0; }
    static constexpr int geom_of(int hj) { return hj & ~1; }      // cv2's half shares the geometry of its group's cv1 half
};

template <int DT>
__device__ __forceinline__ typename Mfma<DT>::frag c3t_frag(const u32x4& p) {
    typename Mfma<DT>::frag f;
    __builtin_memcpy(&f, &p, 16);
    return f;
}

__device__ __forceinline__ f32x16 c3t_bias_acc(const f32x4* bl, int group, int hi) {   // accumulator of cout rows group * 32 .. + 31, initialised with the bias
    f32x16 acc;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const f32x4 b = bl[(group * 4 + gq) * 2 + hi];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[gq * 4 + e] = b[e];
    }
    return acc;
}

template <int DT, int CH>
__global__ __launch_bounds__(512) void c3_tile_kernel(const C3TArgs a, const C3TGeom g) {
    typedef C3TCfg<CH> Cfg;
    constexpr int NP = Cfg::NP, NHJ = Cfg::NHJ, GC = Cfg::GC;
    constexpr int SLOT = 6 * NP * 1024;                       // ring slot = the largest stage (one kernel row of one 32-channel chunk of the 3x3)
    constexpr int NPC_A = 4 * NP, NPC_B = 2 * NP, NPC_C = 6 * NP;   // 1 KiB pieces per stage (phase D = A's size)
    constexpr int PW_A = (NPC_A + 7) / 8, PW_B = (NPC_B + 7) / 8, PW_C = (NPC_C + 7) / 8;   // ... issued per wave (surplus slots re-send the last piece)
    constexpr int BIAS_BYTES = 6 * NP * 32 * 4;               // b12 (2 CH) | bm1 | bm2 | b3 (2 CH)
    constexpr int BG_M1 = 2 * NP, BG_M2 = 3 * NP, BG_3 = 4 * NP;   // 32-cout group index of each bias section (b12 starts at 0)
    typedef typename Mfma<DT>::frag frag;
    static_assert(PW_A <= 3 && PW_B <= 3 && PW_C <= 3, "the counted waits below know 0 .. 3 pieces");
    static_assert(2 * GC + (NHJ & 1) == NHJ, "centre job c = half-jobs (2c, 2c + 1)");

    extern __shared__ __attribute__((aligned(16))) unsigned char c3t_sm[];
    f32x4* const bl = reinterpret_cast<f32x4*>(c3t_sm);
    unsigned char* const T = c3t_sm + BIAS_BYTES;
    const int plane_b = g.nslot * 64;                         // one 32-channel plane of the patch
    unsigned char* const ring = T + NP * plane_b;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, frow = lane & 31;

    const bool has_a = a.nst_a > 0, has_d = a.nst_d > 0;     // block-uniform
    const int S = a.nst_a + 4 * NP + a.nst_d;                 // stages per tile
    const int off_b = a.nst_a * NPC_A * 1024, off_c = off_b + NP * NPC_B * 1024, off_d = off_c + 3 * NP * NPC_C * 1024;

    for (int i = tid; i < BIAS_BYTES / 16; i += 512) bl[i] = *reinterpret_cast<const f32x4*>(a.blob + a.bias_off + i * 16);

    // ---- this wave's half-jobs and the per-lane patch geometry of their slots q = group * 32 + frow = delta + r * pw + c (packed: (r + 8) << 16 | c) ----
    int jg[NHJ], jh[NHJ], jrc[NHJ];
#pragma unroll
    for (int k = 0; k < NHJ; ++k) {
        jg[k] = g.grp[wave][k];
        jh[k] = g.half[wave][k];
        jrc[k] = 0;
        if (Cfg::geom_of(k) != k) continue;
        const int q = (jg[k] >= 0 ? jg[k] : 0) * 32 + frow;
        const int qq = q - g.delta;
        const int qp = qq >= 0 ? qq : 0;
        const int r = fast_div(qp, g.pw, g.magic_pw);
        jrc[k] = (((qq >= 0 ? r : -4) + 8) << 16) | (qp - r * g.pw);   // r = -4: the slots before the first row (never inside); c == w: the pad slot between two rows
    }
    const u32x2 none[4] = {};

    // ---- the weight ring: flat stage sequence of a tile = [A: nst_a][B: NP][C: 3 NP][D: nst_d]; stage ts of the NEXT tile follows the last one ----
    int ts = 0;                 // tile-local index of the stage being consumed
    int cur = 0;                // its ring slot
    int pend = 0;               // pieces this wave issued for the latest stage (0: nothing was issued)
    bool st_pending = false;    // global stores were issued since the last wait: loads and stores retire out of order with each other -> the next wait is vmcnt(0)
    auto issue_stage = [&](int s_idx, int slot) -> int {   // returns the number of pieces this wave issued
        unsigned char* const dst = ring + slot * SLOT;
        const unsigned char* src;
        auto send = [&](int npieces, auto pwt) {
            constexpr int pw_n = decltype(pwt)::value;
#pragma unroll
            for (int j = 0; j < pw_n; ++j) {
                int p = wave + 8 * j;
                p = p < npieces ? p : npieces - 1;
                glds16(reinterpret_cast<const uint16_t*>(src + p * 1024 + lane * 16), reinterpret_cast<uint16_t*>(dst + p * 1024));
            }
        };
        if (s_idx < a.nst_a) {
            src = a.blob + s_idx * (NPC_A * 1024);
            send(NPC_A, std::integral_constant<int, PW_A>{});
            return PW_A;
        }
        s_idx -= a.nst_a;
        if (s_idx < NP) {
            src = a.blob + off_b + s_idx * (NPC_B * 1024);
            send(NPC_B, std::integral_constant<int, PW_B>{});
            return PW_B;
        }
        s_idx -= NP;
        if (s_idx < 3 * NP) {
            src = a.blob + off_c + s_idx * (NPC_C * 1024);
            send(NPC_C, std::integral_constant<int, PW_C>{});
            return PW_C;
        }
        s_idx -= 3 * NP;
        src = a.blob + off_d + s_idx * (NPC_A * 1024);
        send(NPC_A, std::integral_constant<int, PW_A>{});
        return PW_A;
    };
    // top of a step: this wave's pieces of the current stage have landed (everything it issued before the latest stage's pieces), then every wave's
    auto step_wait = [&]() {
        if (st_pending || pend == 0) wait_vmcnt<0>();
        else if (pend == 1) wait_vmcnt<1>();
        else if (pend == 2) wait_vmcnt<2>();
        else wait_vmcnt<3>();
        st_pending = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads of the previous stage / writes of the patch are done
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // the stage two ahead goes into the slot the previous step has just released (every wave is past this step's barrier) -- the LAST vector-memory operation of a step
    auto step_issue = [&](bool more_tiles) {
        int nts = ts + 2;
        bool ok = true;
        if (nts >= S) {
            nts -= S;
            ok = more_tiles;
        }
        const int slot = cur == 0 ? 2 : cur - 1;   // (cur + 2) % 3
        pend = ok ? issue_stage(nts, slot) : 0;
    };
    auto step_done = [&]() {
        ++ts;
        cur = cur == 2 ? 0 : cur + 1;
    };

    // per-tile pixel geometry of the half-jobs, packed: pixel index (clamped into the image) * 4 + bit 0 (inside the image; else: zero padding) + bit 1 (an output pixel of this tile)
    int pmf[NHJ];
    auto tile_geom = [&](int t, int (&pmf_)[NHJ]) {
        const int img = t / g.tiles_per_img, ty = t - img * g.tiles_per_img;
#pragma unroll
        for (int k = 0; k < NHJ; ++k) {
            pmf_[k] = 0;
            if (Cfg::geom_of(k) != k) continue;
            const int r = (jrc[k] >> 16) - 8, c = jrc[k] & 0xffff;
            const int iy = ty * g.R + r - 1;
            const bool slot_ok = jg[k] >= 0 && r >= 0 && r <= g.R + 1 && c < a.w;
            const bool inside = slot_ok && iy >= 0 && iy < a.h;
            const int cy = iy < 0 ? 0 : (iy < a.h ? iy : a.h - 1), cx = c < a.w ? c : a.w - 1;
            pmf_[k] = (((img * a.h + cy) * a.w + cx) << 2) | (inside ? 1 : 0) | ((inside && r >= 1 && r <= g.R) ? 2 : 0);
        }
    };
    // phase A's activation fragments of one k32 stage: lane (pixel, hi) reads channels 32 j + 16 s + 8 hi .. + 7 of its pixel (the cv1-half job of a group loads, its cv2 half shares)
    frag xn[NHJ][2];
#pragma unroll
    for (int k = 0; k < NHJ; ++k)
#pragma unroll
        for (int s = 0; s < 2; ++s) xn[k][s] = frag{};
    auto load_x = [&](int j, const int (&pmf_)[NHJ]) {
#pragma unroll
        for (int k = 0; k < NHJ; ++k) {
            if (!Cfg::maybe_w1(k)) continue;
            if (jg[k] >= 0 && jh[k] == 0) {   // wave-uniform
                const uint16_t* px = a.x + (int64_t)(pmf_[Cfg::geom_of(k)] >> 2) * a.x_cs + 32 * j + 8 * hi;
#pragma unroll
                for (int s = 0; s < 2; ++s) xn[k][s] = *reinterpret_cast<const frag*>(px + 16 * s);
            }
        }
    };

    const int ntiles = g.ntiles;
    int idx = blockIdx.x;
    if (idx >= ntiles) return;
    __syncthreads();   // the biases are in LDS
    // prologue: stages 0 and 1 (and the first tile's first activation fragments between them: the order every later step keeps)
    pend = issue_stage(0, 0);
    if (has_a) {
        tile_geom(xcd_remap(idx, ntiles), pmf);
        load_x(0, pmf);
    }
    pend = issue_stage(1, 1);

    for (; idx < ntiles; idx += gridDim.x) {
        const bool more = idx + (int)gridDim.x < ntiles;
        tile_geom(xcd_remap(idx, ntiles), pmf);
        ts = 0;

        // packets: pk[k] = the rounded outputs of half-job k -- cv1's (k even, or a halo-only job) or cv2's (jh[k] == 1) -- as 16-byte channel octets
        // (cout group i, packet p: octets 2p + hi of the group): exactly the activation fragments of the next 1x1
        u32x4 pk[NHJ][NP][2];
#pragma unroll
        for (int k = 0; k < NHJ; ++k)
#pragma unroll
            for (int i = 0; i < NP; ++i) pk[k][i][0] = pk[k][i][1] = u32x4{0u, 0u, 0u, 0u};

        if (has_a) {
            // ---------------- phase A: cv1 | cv2 over this wave's groups, K = cin ----------------
            f32x16 acc[NHJ][NP];
#pragma unroll
            for (int k = 0; k < NHJ; ++k)
#pragma unroll
                for (int i = 0; i < NP; ++i) acc[k][i] = c3t_bias_acc(bl, (jh[k] == 1 ? NP : 0) + i, hi);
            for (int j = 0; j < a.nst_a; ++j) {
                step_wait();
                frag xc[NHJ][2];
#pragma unroll
                for (int k = 0; k < NHJ; ++k)
#pragma unroll
                    for (int s = 0; s < 2; ++s) xc[k][s] = xn[k][s];
                if (j + 1 < a.nst_a) load_x(j + 1, pmf);
                step_issue(more);
                const unsigned char* const ws = ring + cur * SLOT + lane * 16;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int k = 0; k < NHJ; ++k) {
                        if (jg[k] < 0) continue;   // wave-uniform
                        frag xb = xc[k][s];
                        if (k > 0 && jh[k] == 1) xb = xc[k > 0 ? k - 1 : 0][s];   // cv2's half of the same group
                        const unsigned char* const wk = ws + (jh[k] == 1 ? NP * 2048 : 0);
#pragma unroll
                        for (int i = 0; i < NP; ++i) acc[k][i] = Mfma<DT>::run(*reinterpret_cast<const frag*>(wk + (i * 2 + s) * 1024), xb, acc[k][i]);
                    }
                step_done();
            }
#pragma unroll
            for (int k = 0; k < NHJ; ++k) {
                if (jg[k] < 0) continue;
#pragma unroll
                for (int i = 0; i < NP; ++i) silu_pack_subtile<DT, false, true>(acc[k][i], none, pk[k][i]);
            }
            if (a.mode == 1) {   // HEAD: cv2(x) goes to memory for the TAIL launch
#pragma unroll
                for (int k = 1; k < NHJ; k += 2) {
                    if (jg[k] < 0 || jh[k] != 1) continue;
                    if (pmf[Cfg::geom_of(k)] & 2) {
                        uint16_t* yp = a.y2_out + (int64_t)(pmf[Cfg::geom_of(k)] >> 2) * a.y2_cs + 8 * hi;
#pragma unroll
                        for (int i = 0; i < NP; ++i)
#pragma unroll
                            for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(yp + i * 32 + p * 16) = pk[k][i][p];
                    }
                    st_pending = true;
                }
            }
        } else {
            // MID / TAIL: the Bottleneck's input (and, TAIL, cv2(x)) arrive from memory in packet form
#pragma unroll
            for (int k = 0; k < NHJ; ++k) {
                if (jg[k] < 0) continue;
                if (jh[k] == 1 && !has_d) continue;
                const int64_t m = pmf[Cfg::geom_of(k)] >> 2;
                const uint16_t* src = jh[k] == 1 ? a.y2_in + m * a.y2_cs : a.y1_in + m * a.y1i_cs;
#pragma unroll
                for (int i = 0; i < NP; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p) pk[k][i][p] = *reinterpret_cast<const u32x4*>(src + i * 32 + p * 16 + 8 * hi);
            }
        }

        // ---------------- phase B: t = m.cv1(x1) for every group of this wave, K = CH from the packets; t -> the LDS patch, zero outside the image ----------------
        {
            f32x16 accb[NHJ][NP];
#pragma unroll
            for (int k = 0; k < NHJ; ++k)
#pragma unroll
                for (int i = 0; i < NP; ++i) accb[k][i] = c3t_bias_acc(bl, BG_M1 + i, hi);
            static_for<0, NP>([&](auto jt) {
                constexpr int j = decltype(jt)::value;
                step_wait();
                step_issue(more);
                const unsigned char* const ws = ring + cur * SLOT + lane * 16;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int k = 0; k < NHJ; ++k) {
                        if (!Cfg::maybe_w1(k)) continue;
                        if (jg[k] < 0 || jh[k] != 0) continue;   // wave-uniform
                        const frag xb = c3t_frag<DT>(pk[k][j][s]);
#pragma unroll
                        for (int i = 0; i < NP; ++i) accb[k][i] = Mfma<DT>::run(*reinterpret_cast<const frag*>(ws + (i * 2 + s) * 1024), xb, accb[k][i]);
                    }
                step_done();
            });
#pragma unroll
            for (int k = 0; k < NHJ; ++k) {
                if (!Cfg::maybe_w1(k)) continue;
                if (jg[k] < 0 || jh[k] != 0) continue;
                const int q = jg[k] * 32 + frow;
                unsigned char* const tq = T + q * 64;
                const int swz = (q >> 2) & 3;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    u32x4 o[2];
                    silu_pack_subtile<DT, false, true>(accb[k][i], none, o);
                    if (!(pmf[k] & 1)) o[0] = o[1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(tq + i * plane_b + (((2 * p + hi) ^ swz) * 16)) = o[p];
                }
            }
        }

        // ---------------- phase C: u = m.cv2(t) (3x3) (+ x1) for the centre groups; stages = (32-channel chunk, kernel row), conv_halo8.hip's order ----------------
        bool cj[GC];
#pragma unroll
        for (int c = 0; c < GC; ++c) cj[c] = jg[2 * c + 1] >= 0 && jh[2 * c + 1] == 1;   // wave-uniform
        {
            // LDS byte offsets of the nine taps' fragments (k16 half 0; half 1 = ^ 32) inside a plane.  Slot q' = q + (dy - 1) pw + (dx - 1) holds its four 16-byte channel
            // octets at q' * 64 + ((octet ^ ((q' >> 2) & 3)) * 16): 32 CONSECUTIVE slots under any constant shift cover every 16-byte bank slot once per ds_read_b128 lane
            // group (MI355X_MICROARCH.md, LDS).  Lanes whose slot is no output pixel read a safe slot (results never stored).  Recomputed per tile on purpose (the
            // opaque `fr`): kept across the other phases the 9 GC offsets cost registers the 1x1 phases do not have.
            int fr = frow;
            asm volatile("" : "+v"(fr));
            int ea[GC][9];
#pragma unroll
            for (int c = 0; c < GC; ++c) {
                const int r = (jrc[2 * c] >> 16) - 8, cc = jrc[2 * c] & 0xffff;
                const bool out_px = jg[2 * c] >= 0 && r >= 1 && r <= g.R && cc < a.w;
                const int q = out_px ? jg[2 * c] * 32 + fr : g.delta + g.pw;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int qs = q + (t / 3 - 1) * g.pw + (t % 3 - 1);
                    ea[c][t] = qs * 64 + ((hi ^ ((qs >> 2) & 3)) * 16);
                }
            }
            f32x16 acc[GC][NP];
#pragma unroll
            for (int c = 0; c < GC; ++c)
#pragma unroll
                for (int i = 0; i < NP; ++i) acc[c][i] = c3t_bias_acc(bl, BG_M2 + i, hi);
            for (int j = 0; j < NP; ++j) {
                const unsigned char* const tp = T + j * plane_b;
                static_for<0, 3>([&](auto dyt) {
                    constexpr int dy = decltype(dyt)::value;
                    step_wait();
                    step_issue(more);
                    const unsigned char* const ws = ring + cur * SLOT + lane * 16;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            frag wf[NP];
#pragma unroll
                            for (int i = 0; i < NP; ++i) wf[i] = *reinterpret_cast<const frag*>(ws + ((dx * NP + i) * 2 + s) * 1024);
#pragma unroll
                            for (int c = 0; c < GC; ++c) {
                                if (!cj[c]) continue;
                                const frag tf = *reinterpret_cast<const frag*>(tp + (s ? (ea[c][dy * 3 + dx] ^ 32) : ea[c][dy * 3 + dx]));
#pragma unroll
                                for (int i = 0; i < NP; ++i) acc[c][i] = Mfma<DT>::run(wf[i], tf, acc[c][i]);
                            }
                        }
                    step_done();
                });
            }
#pragma unroll
            for (int c = 0; c < GC; ++c) {
                if (!cj[c]) continue;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    u32x4 o[2];
                    if (a.shortcut) {   // x1 + ...: the shortcut is phase A's packet, un-swapped into accumulator order (conv_common.hpp lean_load_residual)
                        u32x2 rv[4];
                        unswap_residual_packet(pk[2 * c][i][0], rv, 0);
                        unswap_residual_packet(pk[2 * c][i][1], rv, 2);
                        silu_pack_subtile<DT, true, true>(acc[c][i], rv, o);
                    } else {
                        silu_pack_subtile<DT, false, true>(acc[c][i], none, o);
                    }
                    pk[2 * c][i][0] = o[0];
                    pk[2 * c][i][1] = o[1];
                }
                if (!has_d) {   // HEAD / MID: the Bottleneck's output goes to memory
                    if (pmf[2 * c] & 2) {
                        uint16_t* yp = a.y1_out + (int64_t)(pmf[2 * c] >> 2) * a.y1o_cs + 8 * hi;
#pragma unroll
                        for (int i = 0; i < NP; ++i)
#pragma unroll
                            for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(yp + i * 32 + p * 16) = pk[2 * c][i][p];
                    }
                    st_pending = true;
                }
            }
        }

        // ---------------- phase D: y = cv3([u | cv2(x)]), K = 2 CH from the packets ----------------
        if (has_d) {
            f32x16 acc[GC][2 * NP];
#pragma unroll
            for (int c = 0; c < GC; ++c)
#pragma unroll
                for (int i = 0; i < 2 * NP; ++i) acc[c][i] = c3t_bias_acc(bl, BG_3 + i, hi);
            static_for<0, 2 * NP>([&](auto jt) {
                constexpr int j = decltype(jt)::value;
                step_wait();
                if constexpr (j == 2 * NP - 1) {   // the tile's last step: the next tile's first activation fragments go ahead of its stage 1
                    if (more && has_a) {
                        int pmn[NHJ];
                        tile_geom(xcd_remap(idx + (int)gridDim.x, ntiles), pmn);
                        load_x(0, pmn);
                    }
                }
                step_issue(more);
                const unsigned char* const ws = ring + cur * SLOT + lane * 16;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int c = 0; c < GC; ++c) {
                        if (!cj[c]) continue;
                        const frag xb = c3t_frag<DT>(j < NP ? pk[2 * c][j < NP ? j : 0][s] : pk[2 * c + 1][j < NP ? 0 : j - NP][s]);
#pragma unroll
                        for (int i = 0; i < 2 * NP; ++i) acc[c][i] = Mfma<DT>::run(*reinterpret_cast<const frag*>(ws + (i * 2 + s) * 1024), xb, acc[c][i]);
                    }
                step_done();
            });
#pragma unroll
            for (int c = 0; c < GC; ++c) {
                if (!cj[c]) continue;
                uint16_t* yp = a.y + (int64_t)(pmf[2 * c] >> 2) * a.y_cs + 8 * hi;
#pragma unroll
                for (int i = 0; i < 2 * NP; ++i) {
                    u32x4 o[2];
                    silu_pack_subtile<DT, false, true>(acc[c][i], none, o);
                    if (pmf[2 * c] & 2) {
#pragma unroll
                        for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(yp + i * 32 + p * 16) = o[p];
                    }
                }
                st_pending = true;
            }
        } else if (more && has_a) {
            // HEAD: the next tile's first activation fragments (issued after this tile's stores: the next wait is vmcnt(0) anyway)
            int pmn[NHJ];
            tile_geom(xcd_remap(idx + (int)gridDim.x, ntiles), pmn);
            load_x(0, pmn);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// ymi_c3_pack: the four folded weight matrices [rows][k_pad] (ymi_conv_desc.w layout) -> the stage stream.  Piece = one MFMA weight fragment (32 cout rows x 16 k):
// lane l's 16 bytes = row r0 + (l & 31), k = k0 + 8 (l >> 5) .. + 7.
//   A stage j          pieces (group i of [cv1 | cv2] rows, k16 half s) = i * 2 + s:          k0 = 32 j + 16 s
//   B stage j          pieces (group i of m.cv1, s):                                          k0 = 32 j + 16 s
//   C stage 3 j + dy   pieces (dx, group i of m.cv2, s) = (dx * NP + i) * 2 + s:              k0 = (dy * 3 + dx) * CH + 32 j + 16 s
//   D stage j          pieces (group i of cv3, s):                                            k0 = 32 j + 16 s   (k < CH: the Bottleneck's output, then cv2's)
// then the biases as fp32: b12 (2 CH) | bm1 (CH) | bm2 (CH) | b3 (2 CH).
// ---------------------------------------------------------------------------------------------------
struct C3PackArgs {
    const uint16_t *w12, *wm1, *wm2, *w3;
    const float *b12, *bm1, *bm2, *b3;
    int k12, km1, km2, k3;
    int np, ch, nst_a, nst_d;
    int64_t off_b, off_c, off_d, bias_off;
    unsigned char* blob;
};

__global__ void c3_pack_kernel(const C3PackArgs p) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // 16-byte unit
    const int64_t byte = u * 16;
    if (byte >= p.bias_off) {
        const int64_t f = (byte - p.bias_off) / 4;   // first of four floats
        if (f >= 6 * p.ch) return;
        float v[4];
        for (int e = 0; e < 4; ++e) {
            const int64_t i = f + e;
            v[e] = i < 2 * p.ch ? (p.b12 ? p.b12[i] : 0.f) : (i < 3 * p.ch ? p.bm1[i - 2 * p.ch] : (i < 4 * p.ch ? p.bm2[i - 3 * p.ch] : (p.b3 ? p.b3[i - 4 * p.ch] : 0.f)));
        }
        float* o = reinterpret_cast<float*>(p.blob + byte);
        for (int e = 0; e < 4; ++e) o[e] = v[e];
        return;
    }
    const int np = p.np;
    const uint16_t* w;
    int kstride, row0, k0;
    int64_t rel;
    int piece;
    if (byte < p.off_b) {
        rel = byte;
        const int j = (int)(rel / (4 * np * 1024));
        piece = (int)((rel - (int64_t)j * 4 * np * 1024) / 1024);
        w = p.w12; kstride = p.k12; row0 = (piece >> 1) * 32; k0 = 32 * j + 16 * (piece & 1);
    } else if (byte < p.off_c) {
        rel = byte - p.off_b;
        const int j = (int)(rel / (2 * np * 1024));
        piece = (int)((rel - (int64_t)j * 2 * np * 1024) / 1024);
        w = p.wm1; kstride = p.km1; row0 = (piece >> 1) * 32; k0 = 32 * j + 16 * (piece & 1);
    } else if (byte < p.off_d) {
        rel = byte - p.off_c;
        const int st = (int)(rel / (6 * np * 1024));
        piece = (int)((rel - (int64_t)st * 6 * np * 1024) / 1024);
        const int j = st / 3, dy = st - 3 * j;
        const int dx = (piece >> 1) / np, i = (piece >> 1) - dx * np;
        w = p.wm2; kstride = p.km2; row0 = i * 32; k0 = (dy * 3 + dx) * p.ch + 32 * j + 16 * (piece & 1);
    } else {
        rel = byte - p.off_d;
        const int j = (int)(rel / (4 * np * 1024));
        piece = (int)((rel - (int64_t)j * 4 * np * 1024) / 1024);
        w = p.w3; kstride = p.k3; row0 = (piece >> 1) * 32; k0 = 32 * j + 16 * (piece & 1);
    }
    const int l = (int)((rel & 1023) >> 4);
    const u32x4 v = *reinterpret_cast<const u32x4*>(w + (int64_t)(row0 + (l & 31)) * kstride + k0 + 8 * (l >> 5));
    *reinterpret_cast<u32x4*>(p.blob + byte) = v;
}

struct C3TLayout {
    int np, nst_a, nst_d;
    int64_t off_b, off_c, off_d, bias_off, total;
};
static bool c3t_layout(const ymi_c3_desc* d, C3TLayout& L) {
    if (d->c_hidden != 64 && d->c_hidden != 128) return false;
    if (d->mode < 0 || d->mode > 3) return false;
    L.np = d->c_hidden / 32;
    L.nst_a = (d->mode == 0 || d->mode == 1) ? d->c_in / 32 : 0;
    L.nst_d = (d->mode == 0 || d->mode == 3) ? 2 * L.np : 0;
    L.off_b = (int64_t)L.nst_a * 4 * L.np * 1024;
    L.off_c = L.off_b + (int64_t)L.np * 2 * L.np * 1024;
    L.off_d = L.off_c + (int64_t)3 * L.np * 6 * L.np * 1024;
    L.bias_off = L.off_d + (int64_t)L.nst_d * 4 * L.np * 1024;
    L.total = L.bias_off + (int64_t)6 * d->c_hidden * 4;
    return true;
}

static int c3t_check_weights(const ymi_c3_desc* d, const char* who) {
    const int ch = d->c_hidden;
    YMI_REQUIRE(d->dtype == YMI_F16 || d->dtype == YMI_BF16, "%s: 16-bit storage only", who);
    YMI_REQUIRE(ch == 64 || ch == 128, "%s: hidden widths 64 and 128 (got %d)", who, ch);
    YMI_REQUIRE(d->mode >= 0 && d->mode <= 3, "%s: mode %d", who, d->mode);
    YMI_REQUIRE(d->c_out == 2 * ch, "%s: c_out must be 2 * c_hidden (got %d, hidden %d)", who, d->c_out, ch);
    const bool has_a = d->mode == 0 || d->mode == 1, has_d = d->mode == 0 || d->mode == 3;
    if (has_a) YMI_REQUIRE(d->c_in >= 32 && d->c_in % 32 == 0 && d->w12 && d->b12 && d->k12_pad >= d->c_in, "%s: cv1 | cv2 need c_in %% 32 == 0 and packed rows of >= c_in (c_in %d, k_pad %d)", who, d->c_in, d->k12_pad);
    YMI_REQUIRE(d->wm1 && d->bm1 && d->wm2 && d->bm2 && d->km1_pad >= ch && d->km2_pad >= 9 * ch, "%s: Bottleneck weights missing or rows too short", who);
    if (has_d) YMI_REQUIRE(d->w3 && d->b3 && d->k3_pad >= 2 * ch, "%s: cv3 weights missing or rows too short", who);
    YMI_REQUIRE(((d->k12_pad | d->km1_pad | d->km2_pad | d->k3_pad) & 7) == 0, "%s: packed rows must be 16-byte multiples", who);
    return YMI_OK;
}

int64_t c3_blob_bytes(const ymi_c3_desc* d) {
    C3TLayout L;
    if (d == nullptr || !c3t_layout(d, L)) return 0;
    return L.total;
}

int c3_pack_launch(const ymi_c3_desc* d, void* blob, hipStream_t s) {
    YMI_REQUIRE(d != nullptr && blob != nullptr, "ymi_c3_pack: null argument");
    const int rc = c3t_check_weights(d, "ymi_c3_pack");
    if (rc != YMI_OK) return rc;
    C3TLayout L;
    c3t_layout(d, L);
    C3PackArgs p;
    p.w12 = (const uint16_t*)d->w12; p.wm1 = (const uint16_t*)d->wm1; p.wm2 = (const uint16_t*)d->wm2; p.w3 = (const uint16_t*)d->w3;
    p.b12 = d->b12; p.bm1 = d->bm1; p.bm2 = d->bm2; p.b3 = d->b3;
    p.k12 = d->k12_pad; p.km1 = d->km1_pad; p.km2 = d->km2_pad; p.k3 = d->k3_pad;
    p.np = L.np; p.ch = d->c_hidden; p.nst_a = L.nst_a; p.nst_d = L.nst_d;
    p.off_b = L.off_b; p.off_c = L.off_c; p.off_d = L.off_d; p.bias_off = L.bias_off;
    p.blob = (unsigned char*)blob;
    const int64_t units = (L.total + 15) / 16;
    hipLaunchKernelGGL(c3_pack_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, p);
    return check_launch("c3_pack_kernel");
}

// ---- strip geometry: rows per tile R, slot origin delta, the assignment of 32-slot groups to waves ----
// Feasible (R, delta): the centre groups (those holding output pixels) fit the waves' centre jobs, the halo-only groups their spare jobs, the patch fits the LDS.
// Cost model: rounds of tiles over 256 CUs x per-tile time (a centre group ~6x a halo-only group: the 3x3 and cv3 dominate).
template <int CH>
static bool c3t_geometry(int n, int h, int w, C3TGeom& g) {
    typedef C3TCfg<CH> Cfg;
    constexpr int NP = Cfg::NP, GC = Cfg::GC;
    const int pw = w + 1;
    const int lds_max = 160 * 1024;
    const int fixed = 6 * NP * 32 * 4 + 3 * 6 * NP * 1024;
    const int max_groups = (lds_max - fixed) / (NP * 32 * 64);
    double best = 1e30;
    int bR = 0, bD = 0;
    if (const char* e = getenv("YOLORT_AMD_C3T_GEOM")) {   // tuning aid: "R,delta"
        int r_ = 0, d_ = 0;
        if (sscanf(e, "%d,%d", &r_, &d_) == 2) { bR = r_; bD = d_; best = 0; }
    }
    auto eval = [&](int R, int delta, int& ng_t, int& gc0, int& ncen) -> bool {
        const int h_end = delta + (R + 2) * pw;
        ng_t = (h_end + 31) / 32;
        if (ng_t > max_groups || ng_t > 127) return false;
        gc0 = (delta + pw) / 32;
        const int gc1 = (delta + (R + 1) * pw - 2) / 32;
        ncen = gc1 - gc0 + 1;
        const int nhalo = ng_t - ncen;
        if (ncen > 8 * GC) return false;
        if (GC == 1) return nhalo <= 2 * (8 - ncen);
        return nhalo <= 8;
    };
    if (best != 0) {
        for (int R = 1; R <= h; ++R)
            for (int delta = 1; delta <= 32; ++delta) {
                int ng_t, gc0, ncen;
                if (!eval(R, delta, ng_t, gc0, ncen)) continue;
                const int tiles = n * ((h + R - 1) / R);
                const double rounds = (double)((tiles + 255) / 256);
                const int cen_per_wave = (ncen + 7) / 8;
                const double tile_cost = 6.0 * cen_per_wave + 1.0 + 0.02 * ng_t;
                const double cost = rounds * tile_cost * (1.0 + 1e-3 * delta);   // ties: the smaller delta
                if (cost < best) { best = cost; bR = R; bD = delta; }
            }
        if (bR == 0) return false;
    }
    int ng_t, gc0, ncen;
    if (!eval(bR, bD, ng_t, gc0, ncen)) return false;
    memset(&g, 0, sizeof(g));
    g.R = bR; g.pw = pw; g.delta = bD; g.nslot = ng_t * 32;
    g.tiles_per_img = (h + bR - 1) / bR;
    g.ntiles = n * g.tiles_per_img;
    const uint64_t mg = (((uint64_t)1 << 32) / (uint64_t)pw) + 1u;
    g.magic_pw = (unsigned)(mg > 0xffffffffull ? 0xffffffffull : mg);
    for (int wv = 0; wv < 8; ++wv)
        for (int k = 0; k < C3T_MAXHJ; ++k) { g.grp[wv][k] = -1; g.half[wv][k] = 0; }
    // centre groups: GC per wave in order; halo-only groups: the spare jobs, waves with the fewest centre groups first
    int ncw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ncen; ++i) {
        const int wv = (GC == 1) ? i : i / 2, c = (GC == 1) ? 0 : i % 2;
        g.grp[wv][2 * c] = (signed char)(gc0 + i); g.half[wv][2 * c] = 0;
        g.grp[wv][2 * c + 1] = (signed char)(gc0 + i); g.half[wv][2 * c + 1] = 1;
        ++ncw[wv];
    }
    int hq[128], nh = 0;
    for (int gi = 0; gi < ng_t; ++gi)
        if (gi < gc0 || gi >= gc0 + ncen) hq[nh++] = gi;
    int hi_ = 0;
    if (GC == 1) {
        for (int wv = 7; wv >= 0 && hi_ < nh; --wv) {
            if (ncw[wv]) continue;
            for (int k = 0; k < 2 && hi_ < nh; ++k) { g.grp[wv][k] = (signed char)hq[hi_++]; g.half[wv][k] = 0; }
        }
    } else {
        for (int pass = 0; pass <= 2 && hi_ < nh; ++pass)
            for (int wv = 7; wv >= 0 && hi_ < nh; --wv)
                if (ncw[wv] == pass && g.grp[wv][4] < 0) { g.grp[wv][4] = (signed char)hq[hi_++]; g.half[wv][4] = 0; }
    }
    return hi_ == nh;
}

template <int DT, int CH>
static int launch_c3_tile(const C3TArgs& a, const C3TGeom& g, hipStream_t s) {
    constexpr int NP = C3TCfg<CH>::NP;
    const size_t lds = (size_t)6 * NP * 32 * 4 + (size_t)NP * g.nslot * 64 + (size_t)3 * 6 * NP * 1024;
    auto kfn = c3_tile_kernel<DT, CH>;
    if (lds > 64 * 1024) { const int rc = allow_big_lds((const void*)kfn, (int)lds); if (rc != YMI_OK) return rc; }
    int grid = g.ntiles < 256 ? g.ntiles : 256;   // persistent: one block per CU
    if (const char* e = getenv("YOLORT_AMD_C3T_BLOCKS")) { const int v = atoi(e); if (v >= 1 && v < grid) grid = v; }   // tests: few blocks walk many tiles
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, s, a, g);
    return check_launch("c3_tile_kernel");
}

// 1 when ymi_c3_fused takes this descriptor's shape through the strip kernel (the Python side asks before it records the launch)
int c3_tile_supported(const ymi_c3_desc* d) {
    if (d == nullptr || (d->c_hidden != 64 && d->c_hidden != 128) || d->c_out != 2 * d->c_hidden || d->n < 1 || d->h < 1 || d->w < 1) return 0;
    C3TGeom g;
    return (d->c_hidden == 128 ? c3t_geometry<128>(d->n, d->h, d->w, g) : c3t_geometry<64>(d->n, d->h, d->w, g)) ? 1 : 0;
}

int c3_tile_launch(const ymi_c3_desc* d, hipStream_t s) {
    const int rc = c3t_check_weights(d, "ymi_c3_fused");
    if (rc != YMI_OK) return rc;
    YMI_REQUIRE(d->wblob != nullptr, "ymi_c3_fused: hidden width %d needs the fragment-ordered weight stream (ymi_c3_pack -> desc.wblob)", d->c_hidden);
    YMI_REQUIRE(d->n >= 1 && d->h >= 1 && d->w >= 1, "ymi_c3_fused: empty batch");
    const bool has_a = d->mode == 0 || d->mode == 1, has_d = d->mode == 0 || d->mode == 3;
    const int ch = d->c_hidden;
    if (has_a) YMI_REQUIRE(d->x && d->x_cstride % 8 == 0 && d->x_cstride >= d->c_in, "ymi_c3_fused: x must be a 16-byte aligned view of >= c_in channels");
    else YMI_REQUIRE(d->y1_in && d->y1_in_cstride % 8 == 0 && d->y1_in_cstride >= ch, "ymi_c3_fused: modes 2 / 3 read the Bottleneck's input from y1_in");
    if (has_d) YMI_REQUIRE(d->y && d->y_cstride % 8 == 0 && d->y_cstride >= 2 * ch, "ymi_c3_fused: y must be a 16-byte aligned view of >= c_out channels");
    else YMI_REQUIRE(d->y1_out && d->y1_out_cstride % 8 == 0 && d->y1_out_cstride >= ch && d->y1_out != d->y1_in, "ymi_c3_fused: modes 1 / 2 write the Bottleneck's output to y1_out (not in place: neighbouring strips read the halo rows)");
    if (d->mode == 1 || d->mode == 3) YMI_REQUIRE(d->y2 && d->y2_cstride % 8 == 0 && d->y2_cstride >= ch, "ymi_c3_fused: modes 1 / 3 write / read cv2(x) through y2");
    const int64_t cs_max = std::max(std::max((int64_t)d->x_cstride, (int64_t)d->y_cstride), std::max(std::max((int64_t)d->y1_in_cstride, (int64_t)d->y1_out_cstride), (int64_t)d->y2_cstride));
    YMI_REQUIRE((int64_t)d->n * d->h * d->w * cs_max < ((int64_t)1 << 40), "ymi_c3_fused: tensor too large");
    YMI_REQUIRE((int64_t)d->n * d->h * d->w < ((int64_t)1 << 31), "ymi_c3_fused: pixel index must fit 31 bits");
    C3TLayout L;
    c3t_layout(d, L);
    C3TArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const uint16_t*)d->x; a.y1_in = (const uint16_t*)d->y1_in; a.y2_in = (const uint16_t*)d->y2;
    a.y = (uint16_t*)d->y; a.y1_out = (uint16_t*)d->y1_out; a.y2_out = (uint16_t*)d->y2;
    a.blob = (const unsigned char*)d->wblob;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->c_in;
    a.x_cs = d->x_cstride; a.y_cs = d->y_cstride; a.y1i_cs = d->y1_in_cstride; a.y1o_cs = d->y1_out_cstride; a.y2_cs = d->y2_cstride;
    a.mode = d->mode; a.shortcut = d->shortcut ? 1 : 0;
    a.nst_a = L.nst_a; a.nst_d = L.nst_d; a.bias_off = (int)L.bias_off;
    C3TGeom g;
    const bool ok = ch == 128 ? c3t_geometry<128>(d->n, d->h, d->w, g) : c3t_geometry<64>(d->n, d->h, d->w, g);
    YMI_REQUIRE(ok, "ymi_c3_fused: no strip geometry for a %d x %d map at hidden width %d (the halo strip of whole rows must fit the LDS patch)", d->h, d->w, ch);
    if (d->dtype == YMI_F16) return ch == 128 ? launch_c3_tile<YMI_F16, 128>(a, g, s) : launch_c3_tile<YMI_F16, 64>(a, g, s);
    return ch == 128 ? launch_c3_tile<YMI_BF16, 128>(a, g, s) : launch_c3_tile<YMI_BF16, 64>(a, g, s);
}

}  // namespace ymi

extern "C" int64_t ymi_c3_blob_bytes(const ymi_c3_desc* d) { return ymi::c3_blob_bytes(d); }
extern "C" int ymi_c3_pack(const ymi_c3_desc* d, void* blob, void* stream) { return ymi::c3_pack_launch(d, blob, (hipStream_t)stream); }
extern "C" int ymi_c3_tile_supported(const ymi_c3_desc* d) { return ymi::c3_tile_supported(d); }
