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
#include <mutex>
#include <vector>

namespace ymi {


// tuning aid (never in the shipped build): -DC3T_DBG=n removes one ingredient to time the rest (results are garbage): 1 MFMAs, 2 DMA sends, 3 SiLU arithmetic, 4 barriers, 5 fragment reads
#ifndef C3T_DBG
#define C3T_DBG 0
#endif
#ifdef YMI_STAMPS   // tuning aid (never in the shipped build): s_memtime timeline of every wave of the first 256 blocks (tools/stamp_c3t.py); record = id << 56 | cycle
constexpr int C3T_NSTAMP = 256;
__device__ unsigned long long ymi_stamps_c3t[256 * 8 * C3T_NSTAMP];
#define C3T_STAMP(id)                                                                                                                                     \
    do {                                                                                                                                                  \
        if (lane == 0 && blockIdx.x < 256 && st_n < C3T_NSTAMP)                                                                                           \
            ymi_stamps_c3t[(blockIdx.x * 8 + wave) * C3T_NSTAMP + st_n] = ((unsigned long long)(id) << 56) | (__builtin_readcyclecounter() & 0xffffffffffffffull); \
        ++st_n;                                                                                                                                           \
    } while (0)
#else
#define C3T_STAMP(id) ((void)0)
#endif

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
    int nst_a, nst_d;         // k32 chunks of phases A (= cin / 32) / D in this mode (0 when the phase is not part of it)
    int nst_ap;               // ... of phase A in the stream: padded with zero chunks to a multiple of 4 (whole stages, no tail logic in the kernel)
    int bias_off;             // byte offset of the bias section in the blob
};

struct C3TGeom {
    int R, pw, delta, nslot, tiles_per_img, ntiles;
    int ncol, wc, coff;       // column tiles per row band, output columns per tile, the patch column of a tile's first output column: ncol == 1 -> wc = w, pw = w + 1 (ONE shared
                              // pad slot per row), coff = 0; ncol > 1 -> pw = wc + 2 (a halo column either side: the neighbours' pixels, recomputed), coff = 1
    unsigned magic_pw;
    signed char role[8];      // per wave: which static job list it runs (C3TRole)
    signed char grp[8][3];    // per wave: the patch-slot groups of its jobs (-1: none -- the job runs on clamped addresses and writes nothing)
};

// CH = hidden width.  A wave's jobs are 32-slot groups: NC CENTRE groups first (they hold output pixels: cv1 | cv2, m.cv1, the 3x3, cv3), then halo-only groups
// (cv1 and m.cv1 only: the 3x3's input rows above / below the strip).  The job list is STATIC per role so that the MFMA loops are straight-line code.
template <int CH, int ROLE> struct C3TRole;
template <> struct C3TRole<128, 0> { static constexpr int NG = 1, NC = 1; };   // 40 x 40: one centre group ...
template <> struct C3TRole<128, 1> { static constexpr int NG = 2, NC = 0; };   // ... or two halo-only groups
template <> struct C3TRole<64, 0> { static constexpr int NG = 3, NC = 2; };    // 80 x 80: two centre groups and one halo-only group

#if C3T_DBG == 1
#define C3T_MMA(w, x, acc) (acc)
#else
#define C3T_MMA(w, x, acc) Mfma<DT>::run(w, x, acc)
#endif
#if C3T_DBG == 3
#define C3T_SILU false
#else
#define C3T_SILU true
#endif

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

// Explicitly scheduled fragment reads.  hipcc sinks a plain LDS load to just in front of its first use and waits for it with lgkmcnt(0): every MFMA group then
// starts with an exposed LDS round trip (measured on this kernel: SQ_WAIT_ANY 48 % of the wave cycles at 21 % LDS utilisation, profiles/r06e_pmc_c3t.txt).  Here
// the reads are inline assembly (hipcc neither moves nor counts them), issued one whole sub-step ahead into a three-deep register ring, and the wait that releases
// a sub-step's fragments is a counted lgkmcnt that names them as operands (so the MFMAs that use them cannot be scheduled above it).  An outstanding scalar load of
// the compiler's can only make the count stricter.  On the CPU simulator (YMI_HIPSIM) the reads are plain loads and the waits nothing.
__device__ __forceinline__ unsigned c3t_lds_addr(const unsigned char* p) {
#ifdef YMI_HIPSIM
    (void)p;
    return 0u;
#else
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
#endif
}
template <int OFF, class F>
__device__ __forceinline__ void c3t_lds_read(F& dst, const unsigned char* p, unsigned lds_addr) {
#ifdef YMI_HIPSIM
    (void)lds_addr;
    dst = *reinterpret_cast<const F*>(p + OFF);
#else
    (void)p;
#if C3T_DBG == 5
    asm volatile("" : "=v"(dst) : "v"(lds_addr));
#else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF));
#endif
#endif
}
template <int N, class F>
__device__ __forceinline__ void c3t_lds_wait(F& first) {   // at most N LDS reads of this wave still in flight; `first` (and, through c3t_lds_dep, the others) become available here
#ifdef YMI_HIPSIM
    (void)first;
#else
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(first) : "n"(N));
#endif
}
template <class F>
__device__ __forceinline__ void c3t_lds_dep(F& f) {
#ifndef YMI_HIPSIM
    asm volatile("" : "+v"(f));
#else
    (void)f;
#endif
}

__device__ __forceinline__ void c3t_wait_vmcnt(int n) {   // at most n vector-memory operations of this wave still in flight (a smaller count is always safe)
    switch (n) {
#define C3T_VM(k) case k: wait_vmcnt<k>(); break;
        C3T_VM(0) C3T_VM(1) C3T_VM(2) C3T_VM(3) C3T_VM(4) C3T_VM(5) C3T_VM(6) C3T_VM(7) C3T_VM(8) C3T_VM(9) C3T_VM(10) C3T_VM(11) C3T_VM(12) C3T_VM(13) C3T_VM(14) C3T_VM(15)
        C3T_VM(16) C3T_VM(17) C3T_VM(18) C3T_VM(19) C3T_VM(20) C3T_VM(21) C3T_VM(22) C3T_VM(23) C3T_VM(24) C3T_VM(25) C3T_VM(26) C3T_VM(27) C3T_VM(28) C3T_VM(29) C3T_VM(30) C3T_VM(31)
        C3T_VM(32) C3T_VM(33) C3T_VM(34) C3T_VM(35) C3T_VM(36) C3T_VM(37) C3T_VM(38) C3T_VM(39)
#undef C3T_VM
        default: wait_vmcnt<40>(); break;
    }
}

// byte offset of weight fragment (sub-step, cout group i) inside one k32 chunk of [cv1 | cv2] / cv3 rows, and inside one k64 group of m.cv1 rows (see ymi_c3_pack)
template <int NP> struct C3TOffAD { static constexpr int at(int sub, int i) { return (((sub & 1) * NP + i) * 2 + (sub >> 1)) * 1024; } };   // phases A (centre roles) / D: sub = (k16 half s, lower / upper CH rows)
template <int NP> struct C3TOffAH { static constexpr int at(int sub, int i) { return (i * 2 + sub) * 1024; } };                              // phase A, halo-only role: cv1's rows only, sub = s
template <int NP> struct C3TOffB { static constexpr int at(int sub, int i) { return (i * 4 + sub) * 1024; } };                               // phase B: sub = k16 step of the k64 group

template <int DT, int CH, int ROLE>
__device__ __forceinline__ void c3t_run(const C3TArgs& a, const C3TGeom& g, unsigned char* const c3t_sm, const int wave, const int lane) {
    typedef C3TRole<CH, ROLE> Role;
    constexpr int NP = CH / 32, NG = Role::NG, NC = Role::NC, NCA = NC > 0 ? NC : 1;
    constexpr bool XL = CH == 64;   // phase A's activations come through the patch (LDS-DMA) / through a register ring: see below
    // The weight ring: TWO slots of 36 KiB.  A stage is as many whole units of its phase as fit a slot -- few, long steps: every step costs a barrier, a planning
    // pass and an exposed first LDS round trip whatever its length (measured: 34 of 56 us remained with the MFMAs removed at 30 steps per strip, profiles/r06g_dbg.txt)
    //   A / D   unit = one k32 chunk of the 2 CH stacked rows (4 NP KiB);   B   the whole of m.cv1 (2 NP^2 KiB);   C   unit = one tap of one k32 chunk (2 NP KiB)
    constexpr int SLOT = 36 * 1024;
    constexpr int CB = 4 * NP * 1024;                         // bytes of an A / D chunk (= of a k64 group of phase B)
    constexpr int KC = SLOT / CB;                             // chunks per A / D stage: 2 (CH 128) / 4 (CH 64)
    constexpr int NGB = NP / 2;                               // k64 groups of phase B (one stage)
    constexpr int TAPB = 2 * NP * 1024;                       // bytes of a tap
    constexpr int TPS = SLOT / TAPB;                          // taps per C stage: 4 (CH 128) / 9 (CH 64)
    constexpr int NTAP = 9 * NP;                              // taps of the 3x3 in sending order (chunk, dy, dx)
    constexpr int NSTC = NTAP / TPS, NSTD = 2 * NP / KC;      // stages of phases C / D
    static_assert(NTAP % TPS == 0 && (2 * NP) % KC == 0 && NGB * CB <= SLOT, "whole stages");
    constexpr int SZ_A = KC * CB, SZ_B = NGB * CB, SZ_C = TPS * TAPB;
    constexpr int PWMAX = (SLOT / 1024 + 7) / 8;              // pieces a wave sends per stage, at most
    constexpr int BIAS_BYTES = 6 * NP * 32 * 4;               // b12 (2 CH) | bm1 | bm2 | b3 (2 CH)
    constexpr int BG_M1 = 2 * NP, BG_M2 = 3 * NP, BG_3 = 4 * NP;   // 32-cout group index of each bias section (b12 starts at 0)
    typedef typename Mfma<DT>::frag frag;

    const f32x4* const bl = reinterpret_cast<const f32x4*>(c3t_sm);
    unsigned char* const dump = c3t_sm + BIAS_BYTES;          // 1 KiB nobody reads: where a DMA piece that must not land anywhere goes (no branch round it)
    unsigned char* const T = dump + 1024;
    const int plane_b = g.nslot * 64;                         // one 32-channel plane of the patch
    unsigned char* const ring = T + NP * plane_b;
    const int hi = lane >> 5, frow = lane & 31;
    const unsigned l16 = (unsigned)lane * 16u;
#ifdef YMI_STAMPS
    int st_n = 1;
#endif

    const bool has_a = a.nst_a > 0, has_d = a.nst_d > 0;     // block-uniform (nst_a = k32 chunks of phase A)
    const int off_b = a.nst_ap * CB, off_c = off_b + SZ_B, off_d = off_c + NTAP * TAPB;
    const int stream_end = off_d + (has_d ? 2 * NP * CB : 0);

    const int ntiles = g.ntiles;
    int idx = blockIdx.x;
    const int my_tiles = (ntiles - idx + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles this block walks

    // ---- this wave's groups and the per-lane patch geometry of their slots q = group * 32 + frow = delta + r * pw + c (packed: (r + 8) << 16 | c) ----
    int jg[NG], jrc[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        jg[k] = g.grp[wave][k];
        const int q = (jg[k] >= 0 ? jg[k] : 0) * 32 + frow;
        const int qq = q - g.delta;
        const int qp = qq >= 0 ? qq : 0;
        const int r = fast_div(qp, g.pw, g.magic_pw);
        jrc[k] = (((qq >= 0 ? r : -4) + 8) << 16) | (qp - r * g.pw);   // r = -4: the slots before the first row (never inside); c == w: the pad slot between two rows
    }
    const u32x2 none[4] = {};

    // ---- the sender: the stream is contiguous in sending order, so it is a byte cursor that advances by the size of the stage it has just sent (a function of the
    //      section it is in) and wraps to the next tile's first stage; a step sends the stage the NEXT step consumes into the slot the previous step has released ----
    int cur = 0;                // ring slot of the stage being consumed
    int vseq = 0;               // sequence number of the last vector-memory LOAD this wave has issued (pieces, activation fragments, packets); they retire in order
    int pseq = 0;               // ... of the last piece it has sent
    bool st_pending = false;    // global stores were issued since the last wait: loads and stores retire out of order with each other -> the next wait is vmcnt(0)
    int cursor = 0;             // byte offset of the next stage to send
    int sent_slot = 0;          // ring slot it goes to
    constexpr int PW = PWMAX;   // pieces EVERY wave sends per stage (CH 128: every stage is 32 pieces = 4 per wave; CH 64: 8 .. 36 pieces, surplus slots re-send the last piece:
                                // identical bytes) -- no branch: hipcc counts the vector-memory operations in flight only along straight-line code
    auto send_stage = [&]() {
        const int sz = cursor < off_b ? SZ_A : (cursor < off_c ? SZ_B : (cursor < off_d ? SZ_C : SZ_A));
        const unsigned char* const src = a.blob + cursor;
        unsigned char* const dst = ring + sent_slot * SLOT;
        const int np = sz >> 10;
        cursor += sz;
        cursor = cursor >= stream_end ? 0 : cursor;   // (behind the last tile the first stages are sent once more, into slots nobody reads: the kernel drains them before it ends)
        sent_slot ^= 1;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            int p = wave + 8 * j;
            p = p < np ? p : np - 1;
            unsigned lo = l16;
            asm volatile("" : "+v"(lo));   // (opaque: a per-piece lane address hoisted out of the loops costs two registers for the whole kernel -- or a scratch reload in front of the DMA)
#if C3T_DBG != 2
            glds16(reinterpret_cast<const uint16_t*>(src + p * 1024 + lo), reinterpret_cast<uint16_t*>(dst + p * 1024));
#endif
        }
        vseq += PW;
        pseq = vseq;
        asm volatile("" ::: "memory");             // the loads a step issues afterwards stay behind its pieces (the count below relies on the order)
        __builtin_amdgcn_sched_barrier(0);
    };
    // Top of a step.  Vector-memory loads retire in order and the pieces of the stage consumed now were sent during the previous step: they have landed when at most as
    // many loads are in flight as this wave has issued since that send.  Then every wave's (barrier).  The next stage is sent behind the step's first MFMA group: a
    // burst at the head delays the first fragment reads by the pieces' issue time, and where those MFMAs consume operands that came through vector memory (phase A's
    // activation fragments, the packets MID / TAIL read) hipcc waits for them with vmcnt(0) whenever pieces whose number depends on control flow are younger.
    auto step_sync = [&]() {
        C3T_STAMP(1);
        c3t_wait_vmcnt(st_pending ? 0 : vseq - pseq);   // at most the loads issued behind the pieces are still in flight
        st_pending = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads of the previous stage / writes of the patch are done
        C3T_STAMP(2);
#if C3T_DBG != 4
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        C3T_STAMP(3);
    };
    auto step_end = [&]() {
        C3T_STAMP(4);
        cur ^= 1;
    };
    // One stage of a 1x1 GEMM phase: `nch` units (k32 chunks / k64 groups, CB bytes each) of NSUB sub-steps of NP weight fragments (Off::at(sub, i) = byte offset
    // inside the unit), read through a two-deep register ring one sub-step ahead across the units (c3t_lds_read: a buffer is refilled right behind the MFMAs that read
    // it -- its new contents arrive a full LDS round trip, >= 64 cycles, after the refill issues; the MFMAs, issued before it in order, have read their operands by
    // then).  mma(kc, sub, w) issues a sub-step's MFMAs.
    const unsigned ring_lds = c3t_lds_addr(ring);
    auto w_stage = [&](auto nsubt, auto off, auto&& mma) {
        constexpr int NSUB = decltype(nsubt)::value;
        static_assert(NSUB == 2 || NSUB == 4, "even: the buffer of a sub-step does not depend on the unit");
        typedef decltype(off) Off;
        const unsigned char* const ws0 = ring + cur * SLOT + l16;
        const unsigned ws0_lds = ring_lds + (unsigned)(cur * SLOT) + l16;
        const unsigned char *ws = ws0, *wn = ws0;   // this unit's / the next unit's fragments (behind the last unit: the stage's first again -- read, never used)
        unsigned ws_lds = ws0_lds, wn_lds = ws0_lds;
        frag w[2][NP];
        auto rd = [&](auto subt) {   // sub >= NSUB: the next unit's
            constexpr int sub = decltype(subt)::value;
            static_for<0, NP>([&](auto it) {
                constexpr int i = decltype(it)::value;
                if constexpr (sub < NSUB) c3t_lds_read<Off::at(sub, i)>(w[sub % 2][i], ws, ws_lds);
                else c3t_lds_read<Off::at(sub - NSUB, i)>(w[sub % 2][i], wn, wn_lds);
            });
        };
        rd(std::integral_constant<int, 0>{});
        rd(std::integral_constant<int, 1>{});
        static_for<0, KC>([&](auto kct) {   // (units are compile-time: phase A's activation ring is indexed by them)
            constexpr int kc = decltype(kct)::value;
            wn = kc + 1 < KC ? ws + CB : ws0;
            wn_lds = kc + 1 < KC ? ws_lds + CB : ws0_lds;
            static_for<0, NSUB>([&](auto subt) {
                constexpr int sub = decltype(subt)::value;
                // the stage's last two sub-steps read nothing ahead: a read still in flight when the stage ends lands in registers the compiler may have handed to something
                // else (profiles/r06ab5_*: a GPU fault as soon as the wait at the top of the next step was not there to cover it)
                constexpr bool tail = kc == KC - 1 && sub + 2 >= NSUB;
                c3t_lds_wait<(kc == KC - 1 && sub == NSUB - 1) ? 0 : NP>(w[sub % 2][0]);
#pragma unroll
                for (int i = 1; i < NP; ++i) c3t_lds_dep(w[sub % 2][i]);
                mma(kct, subt, w[sub % 2]);
                if constexpr (!tail) rd(std::integral_constant<int, sub + 2>{});
                if constexpr (sub == 0 && kc == 0) send_stage();   // ONE burst behind the first MFMA group (see step_sync)
                __builtin_amdgcn_sched_barrier(0);                 // (the next sub-step's wait names other registers: without the fence hipcc sinks this group's MFMAs below it)
            });
            ws = wn;
            ws_lds = wn_lds;
        });
    };

    // per-tile pixel geometry of the groups, packed: pixel index (clamped into the image) * 4 + bit 0 (inside the image; else: zero padding) + bit 1 (an output pixel of this tile)
    int pmf[NG];
    auto tile_geom = [&](int t, int (&pmf_)[NG]) {
        const int img = t / g.tiles_per_img, rem = t - img * g.tiles_per_img;
        const int ty = rem / g.ncol, c0 = (rem - ty * g.ncol) * g.wc - g.coff;   // image column of patch column 0
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int r = (jrc[k] >> 16) - 8, c = jrc[k] & 0xffff;
            const int iy = ty * g.R + r - 1, ix = c0 + c;
            const bool slot_ok = jg[k] >= 0 && r >= 0 && r <= g.R + 1 && ix >= 0 && ix < a.w;   // (full-width strips: c == w is the pad slot)
            const bool inside = slot_ok && iy >= 0 && iy < a.h;
            const int cy = iy < 0 ? 0 : (iy < a.h ? iy : a.h - 1), cx = ix < 0 ? 0 : (ix < a.w ? ix : a.w - 1);
            const bool out = inside && r >= 1 && r <= g.R && c >= g.coff && c < g.coff + g.wc;
            pmf_[k] = (((img * a.h + cy) * a.w + cx) << 2) | (inside ? 1 : 0) | (out ? 2 : 0);
        }
    };
    // Phase A's activations.  Straight from memory into registers one k32 chunk ahead they bound the phase: ~1.4 us per round trip (HBM / the infinity cache under
    // load) against 0.5 us of MFMAs per chunk, and a deeper register ring spills (profiles/r06i_*, r06q_*).  The patch T is idle until phase B's epilogue, so it is the
    // prefetch ring: plane j % NP holds chunk j of this wave's OWN groups' slots in the patch layout (16-byte octet o of slot q at q * 64 + ((o ^ ((q >> 2) & 3)) * 16)),
    // written by LDS-DMA pieces of 16 slots (lane l: slot l >> 2, position l & 3 -- the swizzle is applied on the source side), NP - 1 chunks ahead, and read back as
    // MFMA fragments exactly like phase C reads t.  No other wave touches these slots before phase C: the only synchronisation is this wave's own counted vmcnt.
    unsigned xqo[NG][2];        // per-lane byte offset of the lane's 16 bytes of chunk 0 for piece p of group k (the launcher checks the tensor stays below 2 GiB)
    int xseq[NP];               // sequence number of the last piece of the chunk in each plane
#pragma unroll
    for (int j = 0; j < NP; ++j) xseq[j] = 0;
    auto x_offsets = [&](int t) {
        const int img = t / g.tiles_per_img, rem = t - img * g.tiles_per_img;
        const int ty = rem / g.ncol, c0 = (rem - ty * g.ncol) * g.wc - g.coff;
#pragma unroll
        for (int k = 0; k < NG; ++k)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int q = (jg[k] >= 0 ? jg[k] : 0) * 32 + p * 16 + (lane >> 2);
                const int qq = q - g.delta;
                const int qp = qq >= 0 ? qq : 0;
                const int r0 = fast_div(qp, g.pw, g.magic_pw);
                const int c = qp - r0 * g.pw;
                const int iy = ty * g.R + (qq >= 0 ? r0 : 0) - 1, ix = c0 + c;
                const int cy = iy < 0 ? 0 : (iy < a.h ? iy : a.h - 1), cx = ix < 0 ? 0 : (ix < a.w ? ix : a.w - 1);   // (slots outside the image read a pixel inside it: masked later)
                const int oct = (lane & 3) ^ ((q >> 2) & 3);
                xqo[k][p] = ((unsigned)((img * a.h + cy) * a.w + cx) * (unsigned)a.x_cs + 8u * (unsigned)oct) * 2u;
            }
    };
    auto issue_x = [&](int jc, int plane, bool real) {   // chunk jc -> `plane`; !real (a refill past the last chunk): into the dump
        const int jcc = jc < a.nst_a ? jc : a.nst_a - 1;  // (the padded chunks multiply a real chunk's finite values by zero weights)
        const char* const xb = reinterpret_cast<const char*>(a.x) + (size_t)jcc * 64;   // block-uniform
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const bool ok = real && jg[k] >= 0;
            unsigned char* const dst = T + plane * plane_b + (jg[k] >= 0 ? jg[k] : 0) * 2048;
#pragma unroll
            for (int p = 0; p < 2; ++p) glds16(reinterpret_cast<const uint16_t*>(xb + xqo[k][p]), reinterpret_cast<uint16_t*>(ok ? dst + p * 1024 : dump));
        }
        vseq += 2 * NG;
        xseq[plane] = real ? vseq : xseq[plane];
        asm volatile("" ::: "memory");
    };
    auto fill_x = [&]() {   // a tile's first NP chunks
#pragma unroll
        for (int j = 0; j < NP; ++j) issue_x(j, j, true);
    };
    // LDS byte offset of this lane's fragment (k16 half 0; half 1 = ^ 32) of its slot in a plane, per group
    int xa[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int q = (jg[k] >= 0 ? jg[k] : 0) * 32 + frow;
        xa[k] = q * 64 + ((hi ^ ((q >> 2) & 3)) * 16);
    }
    const unsigned t_lds0 = c3t_lds_addr(T);
    auto wait_loads = [&](int n) {   // at most n of this wave's loads still in flight (stores in between: loads and stores retire out of order with each other -> all)
        c3t_wait_vmcnt(st_pending ? 0 : n);
        st_pending = false;
        __builtin_amdgcn_wave_barrier();   // (no instruction: what the other lanes' pieces wrote is read below -- the simulator runs lanes apart between such points)
    };

    // Hidden width 128 keeps the earlier form -- straight into a register ring of XF k16 steps per group, slot t % XF reloaded with step t + XF right after its last
    // use (block-uniform base x + 32 t bytes + a 32-bit per-lane byte offset per group): there the patch route measured 4 us SLOWER per launch (its first four
    // chunks, 72 pieces, queue in front of the first MFMA of a block that walks a single tile; profiles/r06r_*), at hidden width 64 9 us faster.
    constexpr int XF = 2;   // (deeper rings were measured slower: with 256 registers the extra slots spill, and every scratch reload is a vmcnt(0) that drains the ring -- profiles/r06i_*)
    frag xf[NG][XF];
#pragma unroll
    for (int k = 0; k < NG; ++k)
#pragma unroll
        for (int t = 0; t < XF; ++t) xf[k][t] = frag{};
    unsigned xo[NG];
    auto xr_offsets = [&](const int (&pmf_)[NG]) {
#pragma unroll
        for (int k = 0; k < NG; ++k) xo[k] = ((unsigned)(pmf_[k] >> 2) * (unsigned)a.x_cs + 8u * (unsigned)hi) * 2u;
    };
    const int nk16 = 2 * a.nst_a;
    auto load_x = [&](int t, auto slott) {   // k16 step t (clamped to the last real one: a surplus reload fetches bytes nobody uses, no branch) into ring slot `slot`
        constexpr int slot = decltype(slott)::value;
        const int tc = t < nk16 ? t : nk16 - 1;
        const char* const xb = reinterpret_cast<const char*>(a.x) + (size_t)tc * 32;   // block-uniform
#pragma unroll
        for (int k = 0; k < NG; ++k) xf[k][slot] = *reinterpret_cast<const frag*>(xb + xo[k]);
        vseq += NG;
    };
    auto load_x_head = [&]() {   // the first XF steps of a tile
        static_for<0, XF>([&](auto tt) { load_x(decltype(tt)::value, tt); });
    };

    // prologue: the first stage, then the first tile's first activation chunks
    send_stage();
    if (has_a) {
        if constexpr (XL) {
            x_offsets(xcd_remap(idx, ntiles));
            fill_x();
        } else {
            tile_geom(xcd_remap(idx, ntiles), pmf);
            xr_offsets(pmf);
            load_x_head();
        }
    }

    for (; idx < ntiles; idx += gridDim.x) {
        const bool more = idx + (int)gridDim.x < ntiles;
        tile_geom(xcd_remap(idx, ntiles), pmf);
        if constexpr (!XL) { if (has_a) xr_offsets(pmf); }
        C3T_STAMP(8);

        // packets: the rounded outputs of a 1x1 as 16-byte channel octets (cout group i, packet p: octets 2p + hi of the group) -- exactly the activation fragments
        // of the next 1x1.  pk1 = cv1's / the Bottleneck's input, later the Bottleneck's output; pk2 = cv2's (centre groups)
        u32x4 pk1[NG][NP][2], pk2[NCA][NP][2];
        // the next tile's first activation chunks go into the patch: once every wave is past its last read of t (behind a barrier that follows phase C)
        auto prefetch_next_x = [&]() {
            if (more && has_a) {
                if constexpr (XL) {
                    x_offsets(xcd_remap(idx + (int)gridDim.x, ntiles));
                    fill_x();
                } else {
                    int pmn[NG];
                    tile_geom(xcd_remap(idx + (int)gridDim.x, ntiles), pmn);
                    xr_offsets(pmn);
                    load_x_head();
                }
            }
        };
        auto head_prefetch = [&]() {   // a launch without phase D (HEAD): one more barrier behind phase C, ahead of the epilogue that hides the round trip
            if constexpr (XL) {
                if (more && has_a) {   // block-uniform
                    __builtin_amdgcn_s_barrier();
                    prefetch_next_x();
                }
            } else {
                prefetch_next_x();
            }
        };

        if (has_a) {
            // ---------------- phase A: cv1 | cv2 over this wave's groups, K = cin; sub-step = (k16 half s, cv1 / cv2 rows): NP weight fragments ----------------
            f32x16 acc1[NG][NP], acc2[NCA][NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
#pragma unroll
                for (int k = 0; k < NG; ++k) acc1[k][i] = c3t_bias_acc(bl, i, hi);
#pragma unroll
                for (int k = 0; k < NCA; ++k) acc2[k][i] = c3t_bias_acc(bl, NP + i, hi);
            }
            if constexpr (XL) {
                // the stage loop is unrolled by XU stages = NP chunks: a chunk's plane is compile-time (the stream pads phase A with zero chunks to multiples of 4)
                constexpr int XU = NP > KC ? NP / KC : 1;
                static_assert((XU * KC) % NP == 0 && 4 % (XU * KC) == 0, "a trip of the unrolled loop starts at plane 0");
                frag xq[NG][2];   // the activation fragments of k16 step t in xq[.][t & 1]: read one step ahead
                wait_loads(vseq - xseq[0]);
#pragma unroll
                for (int k = 0; k < NG; ++k) c3t_lds_read<0>(xq[k][0], T + xa[k], t_lds0 + (unsigned)xa[k]);
                for (int j0 = 0; j0 < a.nst_ap; j0 += XU * KC) {
                    static_for<0, XU>([&](auto ut) {
                        constexpr int u = decltype(ut)::value;
                        const int jb = j0 + u * KC;   // first chunk of this stage
                        step_sync();
                        auto mma = [&](auto kct, auto subt, frag (&w)[NP]) {
                            constexpr int kc = decltype(kct)::value, sub = decltype(subt)::value;
                            constexpr int s = NC > 0 ? sub >> 1 : sub, half = NC > 0 ? sub & 1 : 0;
                            constexpr int plane = (u * KC + kc) % NP;
                            if constexpr (half == 0) {
#pragma unroll
                                for (int k = 0; k < NG; ++k) c3t_lds_dep(xq[k][s]);
#pragma unroll
                                for (int k = 0; k < NG; ++k)
#pragma unroll
                                    for (int i = 0; i < NP; ++i) acc1[k][i] = C3T_MMA(w[i], xq[k][s], acc1[k][i]);
                                // the next k16 step's fragments (ahead of the weight reads w_stage issues next: LDS returns in order, the next wait covers them)
                                if constexpr (s == 0) {
#pragma unroll
                                    for (int k = 0; k < NG; ++k) {
                                        const int e = plane * plane_b + (xa[k] ^ 32);
                                        c3t_lds_read<0>(xq[k][1], T + e, t_lds0 + (unsigned)e);
                                    }
                                } else {
                                    constexpr int pn = (plane + 1) % NP;
                                    wait_loads(vseq - xseq[pn]);   // the next chunk has landed (behind the last chunk: a stale plane, read and never used)
#pragma unroll
                                    for (int k = 0; k < NG; ++k) {
                                        const int e = pn * plane_b + xa[k];
                                        c3t_lds_read<0>(xq[k][0], T + e, t_lds0 + (unsigned)e);
                                    }
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < NC; ++k)
#pragma unroll
                                    for (int i = 0; i < NP; ++i) acc2[k][i] = C3T_MMA(w[i], xq[k][s], acc2[k][i]);
                            }
                            if constexpr ((half == 1 || NC == 0) && s == 1) {   // the chunk's last sub-step: its plane takes the chunk NP ahead
                                const int jn = jb + kc + NP;
                                issue_x(jn, plane, jn < a.nst_a);
                            }
                        };
                        if constexpr (NC > 0) w_stage(std::integral_constant<int, 4>{}, C3TOffAD<NP>{}, mma);
                        else w_stage(std::integral_constant<int, 2>{}, C3TOffAH<NP>{}, mma);
                        step_end();
                    });
                }
            } else {
                constexpr int XU = (XF / 2 + KC - 1) / KC;   // stages per trip round the register ring (the stage loop is unrolled by it: ring slots are compile-time)
                static_assert((XU * KC * 2) % XF == 0 && 4 % (XU * KC) == 0, "a stage starts at a fixed ring slot; the stream pads phase A to multiples of 4 chunks");
                for (int j0 = 0; j0 < a.nst_ap; j0 += XU * KC) {   // (the stream pads phase A with zero chunks to whole stages; their activation loads are clamped to the last real k16 step)
                    static_for<0, XU>([&](auto ut) {
                        constexpr int u = decltype(ut)::value;
                        const int jb = j0 + u * KC;   // first chunk of this stage
                        step_sync();
                        auto mma = [&](auto kct, auto subt, frag (&w)[NP]) {
                            constexpr int kc = decltype(kct)::value, sub = decltype(subt)::value;
                            constexpr int s = NC > 0 ? sub >> 1 : sub, half = NC > 0 ? sub & 1 : 0;
                            constexpr int slot = (2 * (u * KC + kc) + s) % XF;   // ring slot of k16 step 2 (jb + kc) + s
                            if constexpr (half == 0) {
#pragma unroll
                                for (int k = 0; k < NG; ++k)
#pragma unroll
                                    for (int i = 0; i < NP; ++i) acc1[k][i] = C3T_MMA(w[i], xf[k][slot], acc1[k][i]);
                            } else {
#pragma unroll
                                for (int k = 0; k < NC; ++k)
#pragma unroll
                                    for (int i = 0; i < NP; ++i) acc2[k][i] = C3T_MMA(w[i], xf[k][slot], acc2[k][i]);
                            }
                            if constexpr (half == 1 || NC == 0) load_x(2 * (jb + kc) + s + XF, std::integral_constant<int, slot>{});   // the last use of this slot: XF steps ahead
                        };
                        if constexpr (NC > 0) w_stage(std::integral_constant<int, 4>{}, C3TOffAD<NP>{}, mma);
                        else w_stage(std::integral_constant<int, 2>{}, C3TOffAH<NP>{}, mma);
                        step_end();
                    });
                }
                pseq = vseq;   // (the surplus reloads behind the last chunk are dead and may not have been issued at all: do not count on them -- the next wait is vmcnt(0))
#pragma unroll
                for (int k = 0; k < NG; ++k)
#pragma unroll
                    for (int t = 0; t < XF; ++t) xf[k][t] = frag{};   // (the ring is dead until the next tile's head loads: say so, or its registers stay reserved through phases B - D)
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
#pragma unroll
                for (int k = 0; k < NG; ++k) silu_pack_subtile<DT, false, C3T_SILU>(acc1[k][i], none, pk1[k][i]);
#pragma unroll
                for (int k = 0; k < NC; ++k) silu_pack_subtile<DT, false, C3T_SILU>(acc2[k][i], none, pk2[k][i]);
            }
            if (a.mode == 1) {   // HEAD: cv2(x) goes to memory for the TAIL launch
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    if (pmf[k] & 2) {
                        uint16_t* yp = a.y2_out + (int64_t)(pmf[k] >> 2) * a.y2_cs + 8 * hi;
#pragma unroll
                        for (int i = 0; i < NP; ++i)
#pragma unroll
                            for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(yp + i * 32 + p * 16) = pk2[k][i][p];
                    }
                    st_pending = true;
                }
            }
        } else {
            // MID / TAIL: the Bottleneck's input (and, TAIL, cv2(x)) arrive from memory in packet form
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const uint16_t* src = a.y1_in + (int64_t)(pmf[k] >> 2) * a.y1i_cs + 8 * hi;
#pragma unroll
                for (int i = 0; i < NP; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p) pk1[k][i][p] = *reinterpret_cast<const u32x4*>(src + i * 32 + p * 16);
            }
            if (has_d) {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const uint16_t* src = a.y2_in + (int64_t)(pmf[k] >> 2) * a.y2_cs + 8 * hi;
#pragma unroll
                    for (int i = 0; i < NP; ++i)
#pragma unroll
                        for (int p = 0; p < 2; ++p) pk2[k][i][p] = *reinterpret_cast<const u32x4*>(src + i * 32 + p * 16);
                }
            }
            vseq += NG * NP * 2 + (has_d ? NC * NP * 2 : 0);
        }

        C3T_STAMP(9);
        // ---------------- phase B: t = m.cv1(x1) for every group of this wave, K = CH from the packets (ONE stage); t -> the LDS patch, zero outside the image ----------------
        {
            f32x16 accb[NG][NP];
#pragma unroll
            for (int k = 0; k < NG; ++k)
#pragma unroll
                for (int i = 0; i < NP; ++i) accb[k][i] = c3t_bias_acc(bl, BG_M1 + i, hi);
            step_sync();
            {
                // (the k64 groups are compile-time: the packet index is)
                const unsigned char* ws = ring + cur * SLOT + l16;
                const unsigned ws_lds = ring_lds + (unsigned)(cur * SLOT) + l16;
                frag w[2][NP];
                auto rd = [&](auto subt) {
                    constexpr int sub = decltype(subt)::value;
                    static_for<0, NP>([&](auto it) {
                        constexpr int i = decltype(it)::value;
                        c3t_lds_read<(sub / 4) * CB + C3TOffB<NP>::at(sub % 4, i)>(w[sub % 2][i], ws, ws_lds);
                    });
                };
                rd(std::integral_constant<int, 0>{});
                rd(std::integral_constant<int, 1>{});
                static_for<0, 4 * NGB>([&](auto subt) {
                    constexpr int sub = decltype(subt)::value;
                    constexpr int s4 = sub % 4, j = 2 * (sub / 4) + (s4 >> 1), s = s4 & 1;
                    c3t_lds_wait<(sub + 1 < 4 * NGB ? NP : 0)>(w[sub % 2][0]);
#pragma unroll
                    for (int i = 1; i < NP; ++i) c3t_lds_dep(w[sub % 2][i]);
#pragma unroll
                    for (int k = 0; k < NG; ++k)
#pragma unroll
                        for (int i = 0; i < NP; ++i) accb[k][i] = C3T_MMA(w[sub % 2][i], c3t_frag<DT>(pk1[k][j][s]), accb[k][i]);
                    if constexpr (sub + 2 < 4 * NGB) rd(std::integral_constant<int, sub + 2>{});
                    if constexpr (sub == 0) send_stage();   // behind the first MFMAs: the packets may have come through vector memory (MID / TAIL)
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            step_end();
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                if (jg[k] < 0) continue;   // wave-uniform: a job without a group writes nothing
                const int q = jg[k] * 32 + frow;
                unsigned char* const tq = T + q * 64;
                const int swz = (q >> 2) & 3;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    u32x4 o[2];
                    silu_pack_subtile<DT, false, C3T_SILU>(accb[k][i], none, o);
                    if (!(pmf[k] & 1)) o[0] = o[1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(tq + i * plane_b + (((2 * p + hi) ^ swz) * 16)) = o[p];
                }
            }
        }

        C3T_STAMP(10);
        // ---------------- phase C: u = m.cv2(t) (3x3) (+ x1) for the centre groups; taps in conv_halo8.hip's order (32-channel chunk, dy, dx), k16 halves inside ----------------
        if constexpr (NC == 0) {
            for (int j = 0; j < NSTC; ++j) {
                step_sync();
                send_stage();
                step_end();
            }
            if (!has_d) head_prefetch();
        } else {
            // LDS byte offsets of the nine taps' fragments (k16 half 0; half 1 = ^ 32) inside a plane.  Slot q' = q + (dy - 1) pw + (dx - 1) holds its four 16-byte channel
            // octets at q' * 64 + ((octet ^ ((q' >> 2) & 3)) * 16): 32 CONSECUTIVE slots under any constant shift cover every 16-byte bank slot once per ds_read_b128 lane
            // group (MI355X_MICROARCH.md, LDS).  Lanes whose slot is no output pixel read a safe slot (results never stored).  Recomputed per tile on purpose (the
            // opaque `fr`): kept across the other phases the 9 NC offsets cost registers the 1x1 phases do not have.
            int fr = frow;
            asm volatile("" : "+v"(fr));
            int ea[NC][9];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int r = (jrc[c] >> 16) - 8, cc = jrc[c] & 0xffff;
                const bool out_px = jg[c] >= 0 && r >= 1 && r <= g.R && cc >= g.coff && cc < g.coff + g.wc;
                const int q = out_px ? jg[c] * 32 + fr : g.delta + g.pw;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int qs = q + (t / 3 - 1) * g.pw + (t % 3 - 1);
                    ea[c][t] = qs * 64 + ((hi ^ ((qs >> 2) & 3)) * 16);
                }
            }
            f32x16 acc[NC][NP];
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int i = 0; i < NP; ++i) acc[c][i] = c3t_bias_acc(bl, BG_M2 + i, hi);
            const unsigned t_lds = c3t_lds_addr(T);
            static_for<0, NSTC>([&](auto stt) {
                constexpr int st = decltype(stt)::value;
                step_sync();
                const unsigned char* const ws = ring + cur * SLOT + l16;
                const unsigned ws_lds = ring_lds + (unsigned)(cur * SLOT) + l16;
                constexpr int RPS = NP + NC;      // fragment reads per sub-step
                constexpr int NSUB = 2 * TPS;     // sub-step = (tap of this stage, k16 half)
                frag wf[3][NP], tf[3][NC];        // three-deep: a buffer is refilled one whole MFMA group after its last use
                auto read_frags = [&](auto subt) {
                    constexpr int sub = decltype(subt)::value;
                    constexpr int tl = sub >> 1, s = sub & 1, b = sub % 3;
                    constexpr int tg = st * TPS + tl;            // tap in sending order
                    constexpr int j = tg / 9, t9 = tg % 9;       // its 32-channel chunk (= plane of the patch), its (dy, dx)
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int e = j * plane_b + (s ? (ea[c][t9] ^ 32) : ea[c][t9]);
                        c3t_lds_read<0>(tf[b][c], T + e, t_lds + (unsigned)e);
                    }
                    static_for<0, NP>([&](auto it) {
                        constexpr int i = decltype(it)::value;
                        c3t_lds_read<tl * TAPB + (i * 2 + s) * 1024>(wf[b][i], ws, ws_lds);
                    });
                };
                read_frags(std::integral_constant<int, 0>{});
                read_frags(std::integral_constant<int, 1>{});
                static_for<0, NSUB>([&](auto subt) {
                    constexpr int sub = decltype(subt)::value;
                    constexpr int b = sub % 3;
                    c3t_lds_wait<(sub + 1 < NSUB ? RPS : 0)>(tf[b][0]);   // this sub-step's fragments are in; the next one's may still be in flight
#pragma unroll
                    for (int c = 1; c < NC; ++c) c3t_lds_dep(tf[b][c]);
#pragma unroll
                    for (int i = 0; i < NP; ++i) c3t_lds_dep(wf[b][i]);
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int i = 0; i < NP; ++i) acc[c][i] = C3T_MMA(wf[b][i], tf[b][c], acc[c][i]);
                    if constexpr (sub + 2 < NSUB) read_frags(std::integral_constant<int, sub + 2>{});
                    if constexpr (sub == 0) send_stage();
                    __builtin_amdgcn_sched_barrier(0);   // (the next sub-step's wait names other registers: without the fence hipcc sinks this group's MFMAs below it)
                });
                step_end();
            });
            if (!has_d) head_prefetch();
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    u32x4 o[2];
                    if (a.shortcut) {   // x1 + ...: the shortcut is phase A's packet, un-swapped into accumulator order (conv_common.hpp lean_load_residual)
                        u32x2 rv[4];
                        unswap_residual_packet(pk1[c][i][0], rv, 0);
                        unswap_residual_packet(pk1[c][i][1], rv, 2);
                        silu_pack_subtile<DT, true, C3T_SILU>(acc[c][i], rv, o);
                    } else {
                        silu_pack_subtile<DT, false, C3T_SILU>(acc[c][i], none, o);
                    }
                    pk1[c][i][0] = o[0];
                    pk1[c][i][1] = o[1];
                }
                if (!has_d) {   // HEAD / MID: the Bottleneck's output goes to memory
                    if (pmf[c] & 2) {
                        uint16_t* yp = a.y1_out + (int64_t)(pmf[c] >> 2) * a.y1o_cs + 8 * hi;
#pragma unroll
                        for (int i = 0; i < NP; ++i)
#pragma unroll
                            for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(yp + i * 32 + p * 16) = pk1[c][i][p];
                    }
                    st_pending = true;
                }
            }
        }

        C3T_STAMP(11);
        // ---------------- phase D: y = cv3([u | cv2(x)]), K = 2 CH from the packets; sub-step = (k16 half s, lower / upper half of the couts): NP weight fragments ----------------
        if (has_d) {
            if constexpr (NC == 0) {
                for (int j = 0; j < NSTD; ++j) {
                    step_sync();
                    send_stage();
                    if (j == 0) prefetch_next_x();
                    step_end();
                }
            } else {
                f32x16 acc[NC][2 * NP];
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int i = 0; i < 2 * NP; ++i) acc[c][i] = c3t_bias_acc(bl, BG_3 + i, hi);
                static_for<0, NSTD>([&](auto stt) {
                    constexpr int st = decltype(stt)::value;
                    step_sync();
                    const unsigned char* ws = ring + cur * SLOT + l16;
                    const unsigned ws_lds = ring_lds + (unsigned)(cur * SLOT) + l16;
                    frag w[2][NP];
                    auto rd = [&](auto subt) {
                        constexpr int sub = decltype(subt)::value;
                        static_for<0, NP>([&](auto it) {
                            constexpr int i = decltype(it)::value;
                            c3t_lds_read<(sub / 4) * CB + C3TOffAD<NP>::at(sub % 4, i)>(w[sub % 2][i], ws, ws_lds);
                        });
                    };
                    rd(std::integral_constant<int, 0>{});
                    rd(std::integral_constant<int, 1>{});
                    static_for<0, 4 * KC>([&](auto subt) {
                        constexpr int sub = decltype(subt)::value;
                        constexpr int j = st * KC + sub / 4, s = (sub % 4) >> 1, half = sub & 1;   // k32 chunk of K = [u | cv2(x)], k16 half, lower / upper CH couts
                        c3t_lds_wait<(sub + 1 < 4 * KC ? NP : 0)>(w[sub % 2][0]);
#pragma unroll
                        for (int i = 1; i < NP; ++i) c3t_lds_dep(w[sub % 2][i]);
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const frag xb = c3t_frag<DT>(j < NP ? pk1[c][j < NP ? j : 0][s] : pk2[c][j < NP ? 0 : j - NP][s]);
#pragma unroll
                            for (int i = 0; i < NP; ++i) acc[c][half * NP + i] = C3T_MMA(w[sub % 2][i], xb, acc[c][half * NP + i]);
                        }
                        if constexpr (sub + 2 < 4 * KC) rd(std::integral_constant<int, sub + 2>{});
                        if constexpr (sub == 0) {
                            send_stage();
                            if constexpr (st == 0 && XL) prefetch_next_x();   // every wave is past phase C (this stage's barrier)
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    step_end();
                });
                if constexpr (!XL) prefetch_next_x();   // behind the tile's last MFMA, ahead of its epilogue
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    uint16_t* yp = a.y + (int64_t)(pmf[c] >> 2) * a.y_cs + 8 * hi;
#pragma unroll
                    for (int i = 0; i < 2 * NP; ++i) {
                        u32x4 o[2];
                        silu_pack_subtile<DT, false, C3T_SILU>(acc[c][i], none, o);
                        if (pmf[c] & 2) {
#pragma unroll
                            for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(yp + i * 32 + p * 16) = o[p];
                        }
                    }
                    st_pending = true;
                }
            }
        }
        C3T_STAMP(12);
    }
    wait_vmcnt<0>();   // the pieces sent behind the last tile's last stages must not land in another block's LDS
}

template <int DT, int CH>
__global__ __launch_bounds__(512) void c3_tile_kernel(const C3TArgs a, const C3TGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char c3t_sm[];
    constexpr int BIAS_BYTES = 6 * (CH / 32) * 32 * 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef YMI_STAMPS
    int st_n = 0;
#endif
    C3T_STAMP(0);
    if ((int)blockIdx.x >= g.ntiles) return;
    f32x4* const bl = reinterpret_cast<f32x4*>(c3t_sm);
    for (int i = tid; i < BIAS_BYTES / 16; i += 512) bl[i] = *reinterpret_cast<const f32x4*>(a.blob + a.bias_off + i * 16);
    __syncthreads();   // the biases are in LDS
    if constexpr (CH == 128) {
        if (g.role[wave] == 0) c3t_run<DT, 128, 0>(a, g, c3t_sm, wave, lane);   // wave-uniform; every role runs the same sequence of barriers
        else c3t_run<DT, 128, 1>(a, g, c3t_sm, wave, lane);
    } else {
        c3t_run<DT, 64, 0>(a, g, c3t_sm, wave, lane);
    }
}

// ---------------------------------------------------------------------------------------------------
// ymi_c3_pack: the four folded weight matrices [rows][k_pad] (ymi_conv_desc.w layout) -> the stage stream.  Piece = one MFMA weight fragment (32 cout rows x 16 k):
// lane l's 16 bytes = row r0 + (l & 31), k = k0 + 8 (l >> 5) .. + 7.
//   A stage j          pieces (group i of [cv1 | cv2] rows, k16 half s) = i * 2 + s:          k0 = 32 j + 16 s
//   B stage j          pieces (group i of m.cv1, k16 step s of the stage's 64 k) = i * 4 + s:  k0 = 64 j + 16 s
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
        if (j >= p.nst_a) {   // padding chunk of phase A: zero weights
            *reinterpret_cast<u32x4*>(p.blob + byte) = u32x4{0u, 0u, 0u, 0u};
            return;
        }
        piece = (int)((rel - (int64_t)j * 4 * np * 1024) / 1024);
        w = p.w12; kstride = p.k12; row0 = (piece >> 1) * 32; k0 = 32 * j + 16 * (piece & 1);
    } else if (byte < p.off_c) {
        rel = byte - p.off_b;
        const int j = (int)(rel / (4 * np * 1024));
        piece = (int)((rel - (int64_t)j * 4 * np * 1024) / 1024);
        w = p.wm1; kstride = p.km1; row0 = (piece >> 2) * 32; k0 = 64 * j + 16 * (piece & 3);
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
    int np, nst_a, nst_ap, nst_d;
    int64_t off_b, off_c, off_d, bias_off, total;
};
static bool c3t_layout(const ymi_c3_desc* d, C3TLayout& L) {
    if (d->c_hidden != 64 && d->c_hidden != 128) return false;
    if (d->mode < 0 || d->mode > 3) return false;
    L.np = d->c_hidden / 32;
    L.nst_a = (d->mode == 0 || d->mode == 1) ? d->c_in / 32 : 0;
    L.nst_ap = (L.nst_a + 3) / 4 * 4;
    L.nst_d = (d->mode == 0 || d->mode == 3) ? 2 * L.np : 0;
    L.off_b = (int64_t)L.nst_ap * 4 * L.np * 1024;
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
static bool c3t_geometry_search(int n, int h, int w, C3TGeom& g) {
    constexpr int NP = CH / 32, GC = CH == 128 ? 1 : 2;   // centre groups per wave
    const int lds_max = 160 * 1024;
    const int fixed = 6 * NP * 32 * 4 + 1024 + 2 * 36 * 1024;   // biases, the dump, the weight ring
    const int max_groups = (lds_max - fixed) / (NP * 32 * 64);
    double best = 1e30;
    int bR = 0, bD = 0, bN = 0;
    if (const char* e = getenv("YOLORT_AMD_C3T_GEOM")) {   // tuning aid: "R,delta[,column tiles]"
        int r_ = 0, d_ = 0, n_ = 1;
        const int got = sscanf(e, "%d,%d,%d", &r_, &d_, &n_);
        if (got >= 2 && n_ >= 1) { bR = r_; bD = d_; bN = n_; best = 0; }
    }
    // column tiles: ncol == 1 is a full-width strip (pw = w + 1: the rows share one pad slot); wider maps than the patch holds are cut into ncol tiles of wc output columns with
    // a halo column either side (pw = wc + 2)
    auto pw_of = [&](int ncol) { return ncol == 1 ? w + 1 : (w + ncol - 1) / ncol + 2; };
    auto eval = [&](int R, int delta, int ncol, int& ng_t, int& gc0, int& ncen) -> bool {
        const int pw = pw_of(ncol);
        if (ncol > 1 && ((w + ncol - 1) / ncol) * (ncol - 1) >= w) return false;   // (the last column tile would be empty)
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
        for (int ncol = 1; ncol <= 32 && ncol <= w; ++ncol)
            for (int R = 1; R <= h; ++R)
                for (int delta = 1; delta <= 32; ++delta) {
                    int ng_t, gc0, ncen;
                    if (!eval(R, delta, ncol, ng_t, gc0, ncen)) continue;
                    const int tiles = n * ((h + R - 1) / R) * ncol;
                    const double rounds = (double)((tiles + 255) / 256);
                    const int cen_per_wave = (ncen + 7) / 8;
                    const double tile_cost = 6.0 * cen_per_wave + 1.0 + 0.02 * ng_t;
                    const double cost = rounds * tile_cost * (1.0 + 1e-3 * delta) * (ncol > 1 ? 1.05 : 1.0);   // ties: the smaller delta; full-width strips where they fit
                    if (cost < best) { best = cost; bR = R; bD = delta; bN = ncol; }
                }
        if (bR == 0) return false;
    }
    int ng_t, gc0, ncen;
    if (!eval(bR, bD, bN, ng_t, gc0, ncen)) return false;
    const int pw = pw_of(bN);
    memset(&g, 0, sizeof(g));
    g.R = bR; g.pw = pw; g.delta = bD; g.nslot = ng_t * 32;
    g.ncol = bN; g.wc = bN == 1 ? w : (w + bN - 1) / bN; g.coff = bN == 1 ? 0 : 1;
    g.tiles_per_img = ((h + bR - 1) / bR) * bN;
    g.ntiles = n * g.tiles_per_img;
    const uint64_t mg = (((uint64_t)1 << 32) / (uint64_t)pw) + 1u;
    g.magic_pw = (unsigned)(mg > 0xffffffffull ? 0xffffffffull : mg);
    for (int wv = 0; wv < 8; ++wv) {
        g.role[wv] = 0;
        for (int k = 0; k < 3; ++k) g.grp[wv][k] = -1;
    }
    // centre groups: GC per wave in order; halo-only groups: the spare jobs (hidden width 128: the waves without a centre group take two each, role 1;
    // hidden width 64: every wave has one halo-only job, the waves with the fewest centre groups first)
    int ncw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ncen; ++i) {
        const int wv = (GC == 1) ? i : i / 2, c = (GC == 1) ? 0 : i % 2;
        g.grp[wv][c] = (signed char)(gc0 + i);
        ++ncw[wv];
    }
    int hq[128], nh = 0;
    for (int gi = 0; gi < ng_t; ++gi)
        if (gi < gc0 || gi >= gc0 + ncen) hq[nh++] = gi;
    int hi_ = 0;
    if (GC == 1) {
        for (int wv = 7; wv >= 0; --wv) {
            if (ncw[wv]) continue;
            g.role[wv] = 1;
            for (int k = 0; k < 2 && hi_ < nh; ++k) g.grp[wv][k] = (signed char)hq[hi_++];
        }
    } else {
        for (int pass = 0; pass <= 2 && hi_ < nh; ++pass)
            for (int wv = 7; wv >= 0 && hi_ < nh; --wv)
                if (ncw[wv] == pass && g.grp[wv][2] < 0) g.grp[wv][2] = (signed char)hq[hi_++];
    }
    return hi_ == nh;
}

// The search walks ~40 k candidates: once per (n, h, w, hidden width, override), not once per launch (a plan replays a captured graph, but plan building, per-op profiling
// and direct ymi_c3_fused calls launch from the host every time: 70 us of host time per launch before this memo -- profiles/r06ab_*)
template <int CH>
static bool c3t_geometry(int n, int h, int w, C3TGeom& g) {
    struct Entry { int n, h, w; unsigned long env; bool ok; C3TGeom g; };
    static std::mutex mu;
    static std::vector<Entry> memo;
    const char* e = getenv("YOLORT_AMD_C3T_GEOM");
    unsigned long ev = 5381;
    for (const char* p = e; p && *p; ++p) ev = ev * 33 + (unsigned char)*p;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Entry& m : memo)
            if (m.n == n && m.h == h && m.w == w && m.env == ev) { g = m.g; return m.ok; }
    }
    Entry m{n, h, w, ev, false, {}};
    m.ok = c3t_geometry_search<CH>(n, h, w, m.g);
    g = m.g;
    std::lock_guard<std::mutex> lk(mu);
    if (memo.size() >= 64) memo.erase(memo.begin());
    memo.push_back(m);
    return m.ok;
}

template <int DT, int CH>
static int launch_c3_tile(const C3TArgs& a, const C3TGeom& g, hipStream_t s) {
    constexpr int NP = CH / 32;
    const size_t lds = (size_t)6 * NP * 32 * 4 + 1024 + (size_t)NP * g.nslot * 64 + (size_t)2 * 36 * 1024;
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
    YMI_REQUIRE((int64_t)d->n * d->h * d->w < ((int64_t)1 << 29), "ymi_c3_fused: pixel index must fit 29 bits");
    YMI_REQUIRE(!has_a || (int64_t)d->n * d->h * d->w * d->x_cstride < ((int64_t)1 << 30), "ymi_c3_fused: x must stay below 2 GiB (32-bit byte offsets)");
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
    a.nst_a = L.nst_a; a.nst_ap = L.nst_ap; a.nst_d = L.nst_d; a.bias_off = (int)L.bias_off;
    C3TGeom g;
    const bool ok = ch == 128 ? c3t_geometry<128>(d->n, d->h, d->w, g) : c3t_geometry<64>(d->n, d->h, d->w, g);
    YMI_REQUIRE(ok, "ymi_c3_fused: no strip geometry for a %d x %d map at hidden width %d (the halo strip of whole rows must fit the LDS patch)", d->h, d->w, ch);
    if (d->dtype == YMI_F16) return ch == 128 ? launch_c3_tile<YMI_F16, 128>(a, g, s) : launch_c3_tile<YMI_F16, 64>(a, g, s);
    return ch == 128 ? launch_c3_tile<YMI_BF16, 128>(a, g, s) : launch_c3_tile<YMI_BF16, 64>(a, g, s);
}

}  // namespace ymi

#ifdef YMI_STAMPS
extern "C" int ymi_debug_stamps_c3t(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ymi::ymi_stamps_c3t), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
extern "C" int ymi_debug_stamps_c3t_clear(void) {
    static unsigned long long z[256 * 8 * ymi::C3T_NSTAMP];
    return hipMemcpyToSymbol(HIP_SYMBOL(ymi::ymi_stamps_c3t), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int64_t ymi_c3_blob_bytes(const ymi_c3_desc* d) { return ymi::c3_blob_bytes(d); }
extern "C" int ymi_c3_pack(const ymi_c3_desc* d, void* blob, void* stream) { return ymi::c3_pack_launch(d, blob, (hipStream_t)stream); }
extern "C" int ymi_c3_tile_supported(const ymi_c3_desc* d) { return ymi::c3_tile_supported(d); }
extern "C" int ymi_c3_tile_geometry(const ymi_c3_desc* d, int* out6) {
    if (d == nullptr || out6 == nullptr || !ymi::c3_tile_supported(d)) return 0;
    ymi::C3TGeom g;
    const bool ok = d->c_hidden == 128 ? ymi::c3t_geometry<128>(d->n, d->h, d->w, g) : ymi::c3t_geometry<64>(d->n, d->h, d->w, g);
    if (!ok) return 0;
    out6[0] = g.R; out6[1] = g.delta; out6[2] = g.ncol; out6[3] = g.wc; out6[4] = g.nslot; out6[5] = g.ntiles;
    return 1;
}
