// Detection post-process for gfx950: anchor decode + sigmoid + multi-label threshold compaction,
// stable radix sort, class-aware greedy NMS with wave64 ballot/shuffle, top-k + box rescale.
// Integer/index work is bit-exact w.r.t. the oracle contract; float
// arithmetic uses explicitly un-fused fp32 ops so that it follows torch's per-op rounding.
//
// Replaces: yolort/models/box_head.py:328-360 (_concat_pred_logits/_decode_pred_logits),
//           :414-427 (per-image threshold / batched_nms / top-k loop), yolort/models/_utils.py:59-60,
//           yolort/models/anchor_utils.py:19-60 (grids/shifts in closed form),
//           yolort/models/transform.py:354-367 (scale_coords).
//
// Record format (96 bits): hi = img << 32 | ~score_bits (u64), lo = cand = anchor << L | label (u32)
// with L = label bits.  Sorting records ascending by (hi, lo) gives, per image, score descending
// with ties in candidate order (anchor asc, class asc) == a stable descending sort of the
// reference's torch.where order (box_head.py:418).
#include "post_common.hpp"
#include <cstdlib>

namespace ymi {


// ------------------------------------------------------------------------------------------
// 1. decode + threshold.  One wave per feature-map pixel: its 3 x K logits are contiguous in the
//    NHWC fp32 head output, so the wave reads them with coalesced float4 loads (lane l holds
//    channels 4l..4l+3).  Objectness is broadcast with wave shuffles; since cls < 1,
//    score = cls*obj > thr requires obj > thr, so pixels whose three anchors all fail that test
//    skip the class work (the common case).
// ------------------------------------------------------------------------------------------
struct DecodeArgs {
    const float* logits;
    int h, w, cs;          // level geometry, channel stride
    float stride;
    float anc[6];
    int n, K;              // images, outputs per anchor (num_classes + 5)
    int level_off;         // index of this level's first anchor within an image
    CandSink sink;
};

constexpr int DEC_PIX_PER_WAVE = 8;     // pixels walked by one wave
constexpr int DEC_BUF = 512;            // candidate records buffered per wave in LDS before ONE global atomic

__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a) {
    // One wave walks DEC_PIX_PER_WAVE consecutive pixels.  Candidates are appended to a per-wave LDS
    // buffer and flushed with a single atomicAdd on the global counter (one hot word saturates at
    // ~88 atomics/us: one atomic per pixel made this kernel atomic-bound).
    __shared__ uint64_t buf_hi[4][DEC_BUF];
    __shared__ uint32_t buf_lo[4][DEC_BUF];
    const CandSink& k_ = a.sink;
    const int lane = threadIdx.x & 63;
    const int wl = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wl;
    const int64_t npix = (int64_t)a.n * a.h * a.w;
    const int hw = a.h * a.w;
    const int nch = 3 * a.K;
    const int npass = cdiv(nch, 256);
    uint64_t* bhi = buf_hi[wl];
    uint32_t* blo = buf_lo[wl];
    int fill = 0;      // records in the LDS buffer (wave-uniform)
    int fill_img = 0;  // image the buffered records belong to (per-image path flushes on image change)

    auto flush = [&]() {
        flush_records(k_, bhi, blo, fill, fill_img, lane);
        fill = 0;
    };

    for (int pp = 0; pp < DEC_PIX_PER_WAVE; ++pp) {
        const int64_t pix = wave * DEC_PIX_PER_WAVE + pp;
        if (pix >= npix) break;
        const int img = (int)(pix / hw);
        const int rem = (int)(pix - (int64_t)img * hw);
        const int y = rem / a.w, x = rem - y * a.w;
        const float* row = a.logits + pix * a.cs;
        float obj[3];
        bool any_obj = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            obj[k] = sigmoid_acc(row[k * a.K + 4]);
            any_obj |= obj[k] > k_.thr;
        }
        if (lane < 3) {
            const float* r = row + lane * a.K;
            const f32x4 b = decode_box(r[0], r[1], r[2], r[3], x, y, a.stride, a.anc[2 * lane], a.anc[2 * lane + 1]);
            const int anchor = a.level_off + (lane * a.h + y) * a.w + x;
            *reinterpret_cast<f32x4*>(k_.boxes_all + ((int64_t)img * k_.total_anchors + anchor) * 4) = b;
        }
        if (!any_obj) continue;
        if (fill + nch > DEC_BUF || img != fill_img) flush();   // a pixel yields at most nch - 15 records
        fill_img = img;
        for (int p = 0; p < npass; ++p) {
            const int c0 = (p * 64 + lane) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c0 + 3 < nch) v = *reinterpret_cast<const f32x4*>(row + c0);
            else
                for (int e = 0; e < 4; ++e)
                    if (c0 + e < nch) v[e] = row[c0 + e];
            float sc[4];
            unsigned cand[4];
            int cnt = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + e;
                const int k = c / a.K, cls = c - k * a.K - 5;
                bool ok = (c < nch) && (cls >= 0);
                float s = 0.f;
                if (ok) {
                    const float o = k == 0 ? obj[0] : (k == 1 ? obj[1] : obj[2]);
                    s = __fmul_rn(sigmoid_acc(v[e]), o);  // box_head.py:357 scores = cls * obj
                    ok = s > k_.thr;                       // box_head.py:418 strict >
                }
                sc[e] = s;
                const int anchor = a.level_off + (k * a.h + y) * a.w + x;
                cand[e] = ((unsigned)anchor << k_.label_bits) | (unsigned)(cls < 0 ? 0 : cls);
                if (!ok) cand[e] = 0xffffffffu;
                cnt += ok ? 1 : 0;
            }
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d, 64);
                if (lane >= d) incl += t;
            }
            const int total = __shfl(incl, 63, 64);
            if (total == 0) continue;
            if (fill + total > DEC_BUF) flush();   // only reachable when nch > DEC_BUF (many classes)
            int pos = fill + incl - cnt;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (cand[e] != 0xffffffffu) {
                    if (pos < DEC_BUF) {
                        bhi[pos] = ((uint64_t)(unsigned)img << 32) | (uint64_t)(~__float_as_uint(sc[e]));
                        blo[pos] = cand[e];
                    }
                    ++pos;
                }
            }
            fill += total;
            if (fill > DEC_BUF) fill = DEC_BUF;   // unreachable for total <= DEC_BUF; guards the buffer
        }
    }
    flush();
}

__global__ void finalize_count_kernel(int* status, int cap) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (status[ST_NCAND] > cap) status[ST_OVERFLOW] = 1;
    }
}

__device__ __forceinline__ int ncand(const int* status, int cap) {
    const int n = status[ST_NCAND];
    return n < cap ? n : cap;
}

// ------------------------------------------------------------------------------------------
// 2. LSD radix sort, 8 bits per pass, stable.  Three kernels per pass (histogram / scan / scatter).
//    The record count lives on the device (no host sync): blocks past the end exit at once.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned digit_of(uint64_t hi, uint32_t lo, int shift) {
    return shift < 32 ? (lo >> shift) & 0xffu : (unsigned)(hi >> (shift - 32)) & 0xffu;
}

__global__ __launch_bounds__(256) void radix_hist_kernel(const uint64_t* hi, const uint32_t* lo, const int* status, int cap, int shift, uint32_t* hist) {
    __shared__ unsigned cnt[256];
    const int n = ncand(status, cap);
    const int base = blockIdx.x * SORT_ITEMS;
    if (base >= n) return;
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const int end = min(n, base + SORT_ITEMS);
    for (int i = base + threadIdx.x; i < end; i += 256) atomicAdd(&cnt[digit_of(hi[i], lo[i], shift)], 1u);
    __syncthreads();
    hist[(int64_t)blockIdx.x * 256 + threadIdx.x] = cnt[threadIdx.x];
}

// One block per digit d (256 blocks): exclusive scan of hist[b][d] over the blocks b of the pass,
// 256 entries per sweep (wave shuffles + LDS), total of the digit left in tot[d].  The scatter
// kernel turns tot[] into digit bases itself (a 256-value LDS scan), so the serial chain of the
// classic single-block scan is gone.
__global__ __launch_bounds__(256) void radix_scan_kernel(const int* status, int cap, uint32_t* hist, uint32_t* tot) {
    __shared__ unsigned wsum[4];
    __shared__ unsigned carry_s;
    const int n = ncand(status, cap);
    const int nblk = cdiv(n, SORT_ITEMS);
    const int d = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += 256) {
        const int b = b0 + t;
        const unsigned c = b < nblk ? hist[(int64_t)b * 256 + d] : 0u;
        unsigned incl = c;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const unsigned v = __shfl_up(incl, s, 64);
            if (lane >= s) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned off = carry_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (b < nblk) hist[(int64_t)b * 256 + d] = off + incl - c;
        __syncthreads();
        if (t == 255) carry_s = off + incl;
        __syncthreads();
    }
    if (t == 0) tot[d] = carry_s;
}

// scatter: wave w of the block owns the contiguous records [base + w*512, +512), 8 rounds of 64.
__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint64_t* hi_in, const uint32_t* lo_in, uint64_t* hi_out, uint32_t* lo_out,
                                                            const int* status, int cap, int shift, const uint32_t* hist, const uint32_t* tot) {
    __shared__ unsigned wave_hist[4][256];
    __shared__ unsigned wave_base[4][256];
    __shared__ unsigned dsum[256];
    const int n = ncand(status, cap);
    const int base = blockIdx.x * SORT_ITEMS;
    if (base >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 1024; i += 256) (&wave_hist[0][0])[i] = 0;
    {   // digit bases: inclusive scan of the 256 digit totals (Hillis-Steele in LDS)
        const int d = threadIdx.x;
        const unsigned mine = tot[d];
        dsum[d] = mine;
        __syncthreads();
        for (int st = 1; st < 256; st <<= 1) {
            const unsigned v = d >= st ? dsum[d - st] : 0u;
            __syncthreads();
            dsum[d] += v;
            __syncthreads();
        }
        dsum[d] -= mine;  // exclusive
    }
    __syncthreads();
    uint64_t rhi[8];
    uint32_t rlo[8];
    unsigned dig[8];
    uint64_t peers[8];
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = base + wave * 512 + r * 64 + lane;
        const bool valid = i < n;
        rhi[r] = valid ? hi_in[i] : 0;
        rlo[r] = valid ? lo_in[i] : 0;
        const unsigned dg = valid ? digit_of(rhi[r], rlo[r], shift) : 0x100u;  // 0x100 = invalid marker
        dig[r] = dg;
        // peer mask: lanes of this round with the same digit
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t bm = __ballot((dg >> b) & 1u);
            m &= ((dg >> b) & 1u) ? bm : ~bm;
        }
        peers[r] = valid ? m : 0ull;
        if (valid && (m & lt) == 0) wave_hist[wave][dg] += __popcll(m);  // leader lane of the peer group
        // (distinct digits -> distinct LDS addresses; rounds are sequential within the wave)
    }
    __syncthreads();
    {
        const int d = threadIdx.x;
        unsigned run = hist[(int64_t)blockIdx.x * 256 + d] + dsum[d];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            wave_base[w][d] = run;
            run += wave_hist[w][d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (dig[r] != 0x100u) {
            const unsigned rank = __popcll(peers[r] & lt);
            const unsigned dst = wave_base[wave][dig[r]] + rank;
            hi_out[dst] = rhi[r];
            lo_out[dst] = rlo[r];
        }
        // make every lane's read of wave_base[..] precede the leaders' update for this round
        __builtin_amdgcn_wave_barrier();
        if (dig[r] != 0x100u && (peers[r] & lt) == 0) wave_base[wave][dig[r]] += __popcll(peers[r]);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// 2b. per-image sort in LDS (the common case: <= 16384 candidates per image).  decode appended each
//     image's records to its own region, so ONE launch (a 1024-thread block per image) replaces the
//     ten three-kernel radix passes: bitonic sort of the unique 64-bit keys (~score << 32 | cand)
//     gives the G order (score descending, ties by candidate index), a second bitonic sort of
//     (label << 32 | rank) gives the per-class P order the NMS walks.  G and P are written compactly
//     (image offsets = prefix of the counts), exactly as the radix path leaves them.
// ------------------------------------------------------------------------------------------
constexpr int IMG_SORT_MAX = 16384;

__device__ __forceinline__ void bitonic_sort_lds(uint64_t* key, int np2) {
    // Compare-exchange steps with distance j < chunk stay inside one wave's contiguous chunk of the array, so they need
    // no block-wide barrier (the wave's own LDS accesses are ordered): with 16 waves on 4096 keys that is 68 of the 78
    // steps -- the sort was barrier-latency bound (~1.5 us per __syncthreads of 16 waves), not work bound.
    // Small arrays (the common case: ~1000 candidates per image at score_thresh 0.25) keep 128-key chunks on the first
    // np2 / 128 waves and leave the others idle: with 16 x 64-key chunks every one of the 55 steps of a 1024-key sort was a
    // 1024-thread barrier (197 us for two sorts); 8 x 128 leaves 6 block-level steps.
    const int nwaves = blockDim.x >> 6;
    const int min_chunk = np2 < 128 ? np2 : 128;
    const int chunk = np2 / nwaves >= min_chunk ? np2 / nwaves : min_chunk;   // keys per wave (power of two)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 2; k <= np2; k <<= 1) {
        int j = k >> 1;
        for (; j > 0 && j >= chunk; j >>= 1) {   // partners in other waves' chunks
            for (int i = threadIdx.x; i < np2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = key[i], y = key[ixj];
                    const bool asc = (i & k) == 0;
                    if ((x > y) == asc) { key[i] = y; key[ixj] = x; }
                }
            }
            __syncthreads();
        }
        if (j > 0) {                              // the remaining distances: this wave sorts within its own chunk
            uint64_t* mine = key + wave * chunk;
            const int base = wave * chunk;
            const bool active = base < np2;       // waves past the array idle (wave-uniform)
            for (; active && j > 0; j >>= 1) {
                for (int t = lane; t < chunk; t += 64) {
                    const int txj = t ^ j;
                    if (txj > t) {
                        const uint64_t x = mine[t], y = mine[txj];
                        const bool asc = ((base + t) & k) == 0;
                        if ((x > y) == asc) { mine[t] = y; mine[txj] = x; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();                      // next phase starts with cross-chunk distances (or the sort is done)
        }
    }
}

// number of keys of the sorted run `run[0..len)` that are < key (keys are unique)
__device__ __forceinline__ int lower_bound_u64(const uint64_t* run, int len, uint64_t key) {
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (run[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Sorts keys[0..n) (HBM, unique 64-bit keys) ascending into dst order: runs of up to `lds_keys` keys are
// bitonic-sorted in LDS and written back; with more than one run every key's final rank is its rank
// in its own run plus its lower_bound in every other run (merge by ranking, fully parallel).
// rank_out[i] receives the final rank of the key stored at position i after the run sort.
__device__ void sort_runs(uint64_t* keys, int n, uint64_t* lds_key, int lds_keys, uint32_t* rank_out) {
    const int nruns = (n + lds_keys - 1) / lds_keys;
    for (int r = 0; r < nruns; ++r) {
        const int r0 = r * lds_keys;
        const int len = n - r0 < lds_keys ? n - r0 : lds_keys;
        int np2 = 64;
        while (np2 < len) np2 <<= 1;
        for (int i = threadIdx.x; i < np2; i += blockDim.x) lds_key[i] = i < len ? keys[r0 + i] : ~0ull;
        __syncthreads();
        bitonic_sort_lds(lds_key, np2);
        for (int i = threadIdx.x; i < len; i += blockDim.x) keys[r0 + i] = lds_key[i];
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i / lds_keys;
        int rank = i - r * lds_keys;
        if (nruns > 1) {
            const uint64_t k = keys[i];
            for (int q = 0; q < nruns; ++q) {
                if (q == r) { rank += 0; continue; }
                const int q0 = q * lds_keys;
                const int qlen = n - q0 < lds_keys ? n - q0 : lds_keys;
                rank += lower_bound_u64(keys + q0, qlen, k);
            }
            // ranks of earlier runs' elements are already counted through lower_bound; add nothing else
        }
        rank_out[i] = (uint32_t)rank;
    }
    __syncthreads();
}

// Score-prefix selection (per-image path).  Greedy NMS decides each box from higher-scored boxes only, and the
// output keeps the K best survivors: if a score-ordered prefix of an image's candidates already yields K survivors,
// candidates behind it cannot change the result.  For an image with more than 1.5 * sel_t records this kernel keeps
// the records of the best linear score bins (1/4096 wide) that together hold >= sel_t records, compacts them to the
// front of the image's region and flags the image as truncated; gather_topk_kernel raises YMI_STATUS_PREFIX_SHORT
// if such an image ends with fewer than K survivors (the host then re-runs with YMI_POST_EXACT_FULL).
constexpr int SEL_BINS = 4096;
constexpr int RANK_MAX = 6144;   // images with at most this many records (after the prefix selection) are sorted by multi-block ranking

__device__ __forceinline__ int score_bin(uint64_t hi) {
    const float s = __uint_as_float(~(uint32_t)hi);
    const int b = (int)((1.0f - s) * (float)SEL_BINS);
    return b < 0 ? 0 : (b >= SEL_BINS ? SEL_BINS - 1 : b);
}

// histogram increment with the saturated case in mind: when every valid lane of the wave holds the SAME bin (thousands of records with equal or
// near-equal scores: the case the refinement below exists for) one lane adds the wave's count -- 64 lanes hammering one LDS address serialise
__device__ __forceinline__ void hist_add(int* hist, int bin, bool valid) {
    const uint64_t mv = __ballot(valid);
    if (mv == 0ull) return;
    const int first = __builtin_ctzll(mv);
    const int b0 = __builtin_amdgcn_readlane(bin, first);
    const uint64_t same = __ballot(valid && bin == b0);
    if (same == mv) {
        if ((int)(threadIdx.x & 63) == first) atomicAdd(&hist[b0], __popcll(same));
    } else if (valid) {
        atomicAdd(&hist[bin], 1);
    }
}

// Multi-block form of the selection's two streaming passes (round 3).  ONE block per image streams the image twice -- histogram, then the
// in-place compaction -- and at 8 images of 218 k records (yolov5l6 at 1280 x 1280) that is 8 CUs moving 35 MB: 0.39 ms of a 0.52 ms post-process.
// For images with regions of >= 2 * SEL_SLICE records the passes are split over slices of SEL_SLICE records:
//   sel_hist_kernel      (slices x images)  LDS histogram of a slice, non-empty bins added to the image's global histogram
//   select_prefix_kernel (images)           the cut from that histogram; unless the boundary bin is fat (refinement: the one-block path above stays
//                                           in charge of the whole image) it only PUBLISHES the cut: mode 1, b*, and the count it implies
//   sel_compact_kernel   (slices x images)  records of bins <= b* appended to the image's staging region (arrays [1], free until scatter_ranks)
//                                           at positions handed out by one atomic per wave -- their order does not matter: every consumer ranks
//                                           by the records' unique keys
//   sel_copyback_kernel                     staging -> the front of the image's own region (<= RANK_MAX records)
// An in-place multi-block compaction would overwrite records of slice 0 that its block may not have read yet.
// A fat boundary bin (saturated scores: the benchmark configurations' synthetic weights put most images there) is refined by the same exact radix selection as
// in the one-block kernel, as a fixed LADDER of six (sel_rhist_kernel, sel_rpick_kernel) pairs -- 11 key bits per level; the launch sequence does not depend on the
// data, levels an image does not need return at once.  C5 (8 images of 218 k records): post-process 0.52 -> 0.23 ms; C3 (64 images): 0.90 -> 0.73 ms, but the extra
// passes cost the pipelined throughput 1 %, so batches of 32 or more images keep the one-block form by default (profiles/r03z12_sel_ladder_ab.txt).
constexpr int SEL_SLICE = 16384;

__global__ __launch_bounds__(1024) void sel_hist_kernel(const uint64_t* in_hi, const int* img_count, int cap_img, int sel_t, int* ghist) {
    __shared__ int hist[SEL_BINS];
    const int img = blockIdx.y;
    const int raw = img_count[img];
    const int n_i = raw < cap_img ? raw : cap_img;
    if (sel_t <= 0 || n_i <= sel_t + sel_t / 2) return;
    const int i0 = blockIdx.x * SEL_SLICE;
    const int i1 = i0 + SEL_SLICE < n_i ? i0 + SEL_SLICE : n_i;
    if (i0 >= i1) return;
    const uint64_t* hi = in_hi + (int64_t)img * cap_img;
    for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int c0 = i0; c0 < i1; c0 += 4 * blockDim.x) {   // (whole waves enter hist_add: its ballots need every lane)
        uint64_t hv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = c0 + u * (int)blockDim.x + (int)threadIdx.x;
            hv[u] = i < i1 ? hi[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool v = c0 + u * (int)blockDim.x + (int)threadIdx.x < i1;
            hist_add(hist, v ? score_bin(hv[u]) : 0, v);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < SEL_BINS; b += blockDim.x)
        if (hist[b] != 0) atomicAdd(&ghist[(int64_t)img * SEL_BINS + b], hist[b]);
}

// per-image state of the multi-block selection: {mode, b*, shift, need, have, prefix lo, prefix hi, -}.  mode 0: nothing to do here (no cut, or the one-block kernel did
// it); 1: plain cut at bin b*; 2: the boundary bin is fat, refinement in progress; 3: refined -- records of b* are taken iff (g >> shift) <= prefix
constexpr int SEL_STATE = 8;
constexpr int SEL_RB = 11, SEL_RBINS = 1 << SEL_RB, SEL_LEVELS = (64 + SEL_RB - 1) / SEL_RB;   // 11 key bits per refinement level: six levels exhaust the 64-bit key

// one refinement level, counting half: digit histogram of the records of bin b* that share the prefix chosen so far (slices x images)
__global__ __launch_bounds__(1024) void sel_rhist_kernel(const uint64_t* in_hi, const uint32_t* in_lo, const int* img_count, int cap_img, const int* sel_state, int* rhist) {
    __shared__ int hist[SEL_RBINS];
    const int img = blockIdx.y;
    const int* st = sel_state + SEL_STATE * img;
    if (st[0] != 2) return;
    const int bstar = st[1], shift = st[2];
    const uint64_t prefix = (uint64_t)(uint32_t)st[5] | ((uint64_t)(uint32_t)st[6] << 32);
    const int width = shift >= SEL_RB ? SEL_RB : shift, nshift = shift - width;
    const int raw = img_count[img];
    const int n_i = raw < cap_img ? raw : cap_img;
    const int i0 = blockIdx.x * SEL_SLICE;
    const int i1 = i0 + SEL_SLICE < n_i ? i0 + SEL_SLICE : n_i;
    if (i0 >= i1) return;
    const uint64_t* hi = in_hi + (int64_t)img * cap_img;
    const uint32_t* lo = in_lo + (int64_t)img * cap_img;
    for (int i = threadIdx.x; i < SEL_RBINS; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int c0 = i0; c0 < i1; c0 += 4 * blockDim.x) {
        uint64_t hv[4];
        uint32_t lv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = c0 + u * (int)blockDim.x + (int)threadIdx.x;
            hv[u] = i < i1 ? hi[i] : 0ull;
            lv[u] = i < i1 ? lo[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bool v = c0 + u * (int)blockDim.x + (int)threadIdx.x < i1 && score_bin(hv[u]) == bstar;
            const uint64_t g = ((uint64_t)(uint32_t)hv[u] << 32) | lv[u];
            v = v && (shift >= 64 || (g >> shift) == prefix);
            hist_add(hist, (int)((g >> nshift) & (uint64_t)((1 << width) - 1)), v);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < SEL_RBINS; b += blockDim.x)
        if (hist[b] != 0) atomicAdd(&rhist[(int64_t)img * SEL_RBINS + b], hist[b]);
}

// ... deciding half (one wave per image): whole digit bins are kept in key order until the target is reached; descend into the bin that crosses it, or stop when the
// selection fits RANK_MAX (the arithmetic of select_prefix_kernel's one-block refinement).  Leaves the image's digit histogram zeroed for the next level.
__global__ __launch_bounds__(64) void sel_rpick_kernel(int* sel_state, int* rhist, int* sel_count) {
    const int img = blockIdx.x, lane = threadIdx.x;
    int* st = sel_state + SEL_STATE * img;
    if (st[0] != 2) return;
    int* h = rhist + (int64_t)img * SEL_RBINS;
    const int shift = st[2], need = st[3], have = st[4];
    const uint64_t prefix = (uint64_t)(uint32_t)st[5] | ((uint64_t)(uint32_t)st[6] << 32);
    const int width = shift >= SEL_RB ? SEL_RB : shift, nshift = shift - width;
    constexpr int PER = SEL_RBINS / 64;
    int sum = 0;
    for (int b = 0; b < PER; ++b) sum += h[lane * PER + b];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    const int excl = incl - sum;
    int digit = 0, before = 0, inbin = 0;
    const bool mine = excl < need && incl >= need;   // exactly one lane
    if (mine) {
        int c = excl, b = lane * PER;
        for (; b < lane * PER + PER; ++b) {
            if (c + h[b] >= need) break;
            c += h[b];
        }
        digit = b;
        before = c;
        inbin = h[b];
    }
    const uint64_t owner = __ballot(mine);
    const int src = owner ? __builtin_ctzll(owner) : 0;
    digit = __shfl(digit, src, 64);
    before = __shfl(before, src, 64);
    inbin = __shfl(inbin, src, 64);
    __builtin_amdgcn_wave_barrier();   // every lane has read its counts before they are cleared
    for (int b = 0; b < PER; ++b) h[lane * PER + b] = 0;
    if (lane == 0) {
        const uint64_t np = (shift < 64 ? (prefix << width) : 0ull) | (uint64_t)digit;
        st[2] = nshift;
        st[5] = (int)(uint32_t)np;
        st[6] = (int)(uint32_t)(np >> 32);
        if (have + before + inbin <= RANK_MAX || nshift == 0) {   // keep the whole crossing digit bin: the selection fits (or the key is exhausted: unique keys)
            st[0] = 3;
            sel_count[img] = have + before + inbin;
        } else {
            st[4] = have + before;   // digit bins below the crossing one are taken whole
            st[3] = need - before;
        }
    }
}

__global__ __launch_bounds__(1024) void sel_compact_kernel(const uint64_t* in_hi, const uint32_t* in_lo, const int* img_count, int cap_img, const int* sel_state, int* gfill,
                                                           uint64_t* st_hi, uint32_t* st_lo) {
    const int img = blockIdx.y;
    const int* st = sel_state + SEL_STATE * img;
    const int mode = st[0];
    if (mode != 1 && mode != 3) return;
    const int bstar = st[1];
    const int sel_shift = mode == 3 ? st[2] : 64;
    const uint64_t sel_prefix = (uint64_t)(uint32_t)st[5] | ((uint64_t)(uint32_t)st[6] << 32);
    const int raw = img_count[img];
    const int n_i = raw < cap_img ? raw : cap_img;
    const int i0 = blockIdx.x * SEL_SLICE;
    const int i1 = i0 + SEL_SLICE < n_i ? i0 + SEL_SLICE : n_i;
    if (i0 >= i1) return;
    const uint64_t* hi = in_hi + (int64_t)img * cap_img;
    const uint32_t* lo = in_lo + (int64_t)img * cap_img;
    uint64_t* sh = st_hi + (int64_t)img * cap_img;
    uint32_t* sl = st_lo + (int64_t)img * cap_img;
    const int lane = threadIdx.x & 63;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int c0 = i0; c0 < i1; c0 += 4 * blockDim.x) {
        uint64_t h[4];
        uint32_t l[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = c0 + u * (int)blockDim.x + (int)threadIdx.x;
            h[u] = i < i1 ? hi[i] : 0ull;
            l[u] = i < i1 ? lo[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = score_bin(h[u]);
            const bool take = c0 + u * (int)blockDim.x + (int)threadIdx.x < i1 &&
                              (b < bstar || (b == bstar && (sel_shift >= 64 || ((((uint64_t)(uint32_t)h[u] << 32) | l[u]) >> sel_shift) <= sel_prefix)));
            const uint64_t m = __ballot(take);
            if (m == 0ull) continue;   // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&gfill[img], __popcll(m));
            base = __shfl(base, 0, 64);
            if (take) {
                const int pos = base + __popcll(m & lt);
                sh[pos] = h[u];
                sl[pos] = l[u];
            }
        }
    }
}

__global__ __launch_bounds__(256) void sel_copyback_kernel(uint64_t* hi0, uint32_t* lo0, const uint64_t* st_hi, const uint32_t* st_lo, const int* sel_state, const int* sel_count,
                                                           int cap_img) {
    const int img = blockIdx.y;
    const int mode = sel_state[SEL_STATE * img];
    if (mode != 1 && mode != 3) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < sel_count[img]) {
        hi0[(int64_t)img * cap_img + i] = st_hi[(int64_t)img * cap_img + i];
        lo0[(int64_t)img * cap_img + i] = st_lo[(int64_t)img * cap_img + i];
    }
}

// ghist / sel_mode non-null: the multi-block form (see above) -- the image's level-0 histogram is already in ghist, and an image whose cut needs no refinement is
// left to sel_compact_kernel
__global__ __launch_bounds__(1024) void select_prefix_kernel(uint64_t* in_hi, uint32_t* in_lo, const int* img_count, int cap_img, int n_img, int sel_t,
                                                             int* sel_count, uint32_t* rank_g, uint32_t* rank_p, const int* ghist, int* sel_mode) {
    __shared__ int hist[SEL_BINS];
    __shared__ int s_bstar, s_nsel, s_fill;
    const int img = blockIdx.x;
    const int raw = img_count[img];
    const int n_i = raw < cap_img ? raw : cap_img;
    if (rank_g != nullptr) {   // rank counters of the multi-block ranking sort (rank_image_kernel adds into them)
        const int nz = n_i < RANK_MAX ? n_i : RANK_MAX;
        for (int i = threadIdx.x; i < nz; i += blockDim.x) { rank_g[(int64_t)img * cap_img + i] = 0u; rank_p[(int64_t)img * cap_img + i] = 0u; }
    }
    if (sel_t <= 0 || n_i <= sel_t + sel_t / 2) {
        if (threadIdx.x == 0) { sel_count[img] = n_i; sel_count[n_img + img] = 0; }
        return;
    }
    uint64_t* hi = in_hi + (int64_t)img * cap_img;
    uint32_t* lo = in_lo + (int64_t)img * cap_img;
    if (ghist != nullptr) {   // wave-uniform: sel_hist_kernel has counted the bins
        for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x) hist[i] = ghist[(int64_t)img * SEL_BINS + i];
    } else {
        for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        // One block streams a whole image (up to cap_img records: 2.6 MB at 218 k): four loads in flight per thread, or the pass is bound by one
        // CU's load latency (a single 8-byte load per thread and iteration streamed ~60 GB/s)
        for (int c0 = 0; c0 < n_i; c0 += 4 * blockDim.x) {   // (whole waves enter hist_add: its ballots need every lane)
            uint64_t hv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = c0 + u * (int)blockDim.x + (int)threadIdx.x;
                hv[u] = i < n_i ? hi[i] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool v = c0 + u * (int)blockDim.x + (int)threadIdx.x < n_i;
                hist_add(hist, v ? score_bin(hv[u]) : 0, v);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // first wave: lane l owns bins [64 l, 64 l + 64)
        const int lane = threadIdx.x;
        int sum = 0;
        for (int b = 0; b < 64; ++b) sum += hist[lane * 64 + b];
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        const int excl = incl - sum;
        if (excl < sel_t && incl >= sel_t) {   // exactly one lane (n_i > sel_t)
            int c = excl, b = lane * 64;
            for (; b < lane * 64 + 64; ++b) {
                c += hist[b];
                if (c >= sel_t) break;
            }
            s_bstar = b;
            s_nsel = c;
        }
        if (lane == 0) s_fill = 0;
    }
    __syncthreads();
    const int bstar = s_bstar, nsel = s_nsel;
    if (nsel >= n_i) {   // the best bins hold everything: nothing to cut
        if (threadIdx.x == 0) { sel_count[img] = n_i; sel_count[n_img + img] = 0; }
        return;
    }
    if (sel_mode != nullptr) {   // multi-block form: publish the cut (plain), or hand the fat boundary bin to the refinement ladder; sel_compact_kernel moves the records
        if (threadIdx.x == 0) {
            int* st = sel_mode + SEL_STATE * img;
            st[1] = bstar;
            sel_count[n_img + img] = 1;
            if (nsel <= RANK_MAX) {
                st[0] = 1;
                sel_count[img] = nsel;
            } else {
                const int before_bin = nsel - hist[bstar];   // records of the better bins
                st[0] = 2;
                st[2] = 64;
                st[3] = sel_t - before_bin;   // still wanted from the boundary bin (>= 1)
                st[4] = before_bin;           // taken for sure
                st[5] = st[6] = 0;
            }
        }
        return;
    }
    // Round 3: a FAT boundary bin (saturated scores: thousands of records within 1/4096 of each other, or exactly equal) used to push the
    // image past RANK_MAX and onto the one-block-per-image sort (C3: 1.0 ms, C5: 0.33 ms of the post-process).  The cut is refined inside
    // that bin by an exact radix selection on the records' full sort key g = ~score << 32 | lo (unique): 11 bits per level from the top, at
    // each level the digit bins of the records that share the chosen prefix are counted, whole digit bins are kept in key order until the
    // target is reached, and the level descends into the bin that crosses it -- until the selection fits RANK_MAX (any set "every record
    // with key <= T" is a score-ordered prefix, so stopping early with a superset is exact).  Level 0 is the linear bin cut above, so
    // images that never had a fat bin are selected exactly as before.
    uint64_t sel_prefix = 0ull;   // records of bin bstar are taken iff (g >> sel_shift) <= sel_prefix
    int sel_shift = 64;           // 64: the whole bin (no refinement)
    if (nsel > RANK_MAX) {
        constexpr int RB = 11, RBINS = 1 << RB;
        __shared__ int s_digit, s_before;
        const int lane0 = threadIdx.x & 63;
        int before_bin = nsel - hist[bstar];          // records of the better bins (hist is intact: read-only since the scan)
        int need = sel_t - before_bin;                // still wanted from the boundary bin (>= 1)
        int have = before_bin;                        // taken for sure so far
        uint64_t prefix = 0ull;
        int shift = 64;
        __syncthreads();
        while (shift > 0) {
            const int width = shift >= RB ? RB : shift;
            const int nshift = shift - width;
            for (int i = threadIdx.x; i < RBINS; i += blockDim.x) hist[i] = 0;   // (SEL_BINS >= RBINS; bin counts of level 0 are no longer needed)
            __syncthreads();
            for (int c0 = 0; c0 < n_i; c0 += 4 * blockDim.x) {
                uint64_t hv[4];
                uint32_t lv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {   // all eight loads of the thread first
                    const int i = c0 + u * (int)blockDim.x + (int)threadIdx.x;
                    hv[u] = i < n_i ? hi[i] : 0ull;
                    lv[u] = i < n_i ? lo[i] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bool v = c0 + u * (int)blockDim.x + (int)threadIdx.x < n_i && score_bin(hv[u]) == bstar;
                    const uint64_t g = ((uint64_t)(uint32_t)hv[u] << 32) | lv[u];
                    v = v && (shift >= 64 || (g >> shift) == prefix);
                    hist_add(hist, (int)((g >> nshift) & (uint64_t)((1 << width) - 1)), v);
                }
            }
            __syncthreads();
            if (threadIdx.x < 64) {   // first wave: lane l owns digits [32 l, 32 l + 32)
                int sum = 0;
                for (int b = 0; b < RBINS / 64; ++b) sum += hist[lane0 * (RBINS / 64) + b];
                int incl = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int t = __shfl_up(incl, d, 64);
                    if (lane0 >= d) incl += t;
                }
                const int excl = incl - sum;
                if (excl < need && incl >= need) {   // exactly one lane
                    int c = excl, b = lane0 * (RBINS / 64);
                    for (; b < lane0 * (RBINS / 64) + RBINS / 64; ++b) {
                        if (c + hist[b] >= need) break;
                        c += hist[b];
                    }
                    s_digit = b;
                    s_before = c;
                }
            }
            __syncthreads();
            const int digit = s_digit, before = s_before, inbin = hist[digit];
            prefix = (shift < 64 ? (prefix << width) : 0ull) | (uint64_t)digit;
            shift = nshift;
            if (have + before + inbin <= RANK_MAX || shift == 0) break;   // keep the whole crossing digit bin: the selection fits (or the key is exhausted: unique keys)
            have += before;          // digit bins below the crossing one are taken whole
            need -= before;
            __syncthreads();
        }
        sel_prefix = prefix;
        sel_shift = shift;
        __syncthreads();
        if (threadIdx.x == 0) s_fill = 0;
        __syncthreads();
    }
    // in-place compaction: a chunk is read completely before any of its survivors is written, and survivors only move to positions at or
    // before indices already read.  Chunks of 8 records per thread (two block barriers per 8192 records: with one record per thread the
    // barriers of this loop were most of the kernel -- 213 iterations for a 218 k-record image)
    const int lane = threadIdx.x & 63;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    constexpr int CU_ = 8;
    for (int c0 = 0; c0 < n_i; c0 += CU_ * blockDim.x) {
        uint64_t h[CU_];
        uint32_t l[CU_];
        bool take[CU_];
#pragma unroll
        for (int u = 0; u < CU_; ++u) {
            const int i = c0 + u * (int)blockDim.x + (int)threadIdx.x;
            h[u] = i < n_i ? hi[i] : 0ull;
            l[u] = i < n_i ? lo[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < CU_; ++u) {
            const int i = c0 + u * (int)blockDim.x + (int)threadIdx.x;
            const int b = score_bin(h[u]);
            take[u] = i < n_i && (b < bstar || (b == bstar && (sel_shift >= 64 || ((((uint64_t)(uint32_t)h[u] << 32) | l[u]) >> sel_shift) <= sel_prefix)));
        }
        __syncthreads();   // the whole chunk has been read
#pragma unroll
        for (int u = 0; u < CU_; ++u) {
            const uint64_t m = __ballot(take[u]);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(&s_fill, __popcll(m));
            base = __shfl(base, 0, 64);
            if (take[u]) {
                const int pos = base + __popcll(m & lt);
                hi[pos] = h[u];
                lo[pos] = l[u];
            }
        }
        __syncthreads();   // its survivors are written (all at positions below c0 + 8192)
    }
    if (threadIdx.x == 0) { sel_count[img] = s_fill; sel_count[n_img + img] = 1; }   // s_fill == nsel when the cut was not refined
}

// ------------------------------------------------------------------------------------------
// 2c. multi-block sort by RANKING (images of <= RANK_MAX records -- every image after the score-prefix selection, which
//     keeps <= 1.5 x 4096).  One 1024-thread block per image made the batch wait for its most crowded image (C2: 32 images,
//     1 k records on average but 5 k in one of them: 194 us for two bitonic sorts of 8192 keys on ONE CU while 255 idled).
//     Keys are unique, so the position of a key is the number of keys below it: block (bi, bj, img) counts, for its 256 keys,
//     the keys of j-slice bj that are smaller -- under BOTH orders at once (G: ~score << 32 | cand; P: label, then G order,
//     i.e. label << (64-L) | G >> L) -- and adds the counts to the keys' rank counters; scatter_ranks_kernel then writes G and
//     P directly in place.  n^2 / 64 wave-iterations of two 64-bit compares: 25 M pairs for 5 k records = ~10 us on a few
//     hundred blocks, and no block-wide barrier in the loop.
// ------------------------------------------------------------------------------------------
constexpr int RANK_IT = 256;    // keys ranked per block
constexpr int RANK_JT = 256;    // keys of a j-slice (4 KiB of LDS: G and P key of each)
constexpr int RANK_GRID = 1024; // blocks of the ranking kernel (they walk the pair list of the whole batch)
constexpr int RANK_NIMG = 1024; // images per batch the ranking kernel's per-block table holds (larger batches: the one-block-per-image sort)

__device__ __forceinline__ uint64_t p_key_of(uint64_t g, int label_bits) {
    return ((g & ((1ull << label_bits) - 1ull)) << (64 - label_bits)) | (g >> label_bits);
}

__global__ __launch_bounds__(RANK_IT) void rank_image_kernel(const uint64_t* in_hi, const uint32_t* in_lo, const int* sel_count, int n_img, int cap_img, int label_bits,
                                                             uint32_t* rank_g, uint32_t* rank_p) {
    // round 6: a fixed grid walks the (256-key tile, 256-key j-slice) pairs of ALL images as one list.  The grid used to be sized for RANK_MAX records per image (24 x 6
    // tiles: on C2 -- 1 k records per image, 5 k in one -- 97 % of 4 608 blocks found nothing to do and the others each scanned 1 024 keys: 42 us); blocks per image
    // fixed at 16 made the crowded image's 400 pairs the tail (124 us).  profiles/r06zz_post_kernels.txt
    __shared__ __attribute__((aligned(16))) uint64_t keys[RANK_JT][2];
    __shared__ int s_cnt[RANK_NIMG];   // records per image (0: none of this kernel's), read ONCE per block (a chain of dependent scalar loads per pair cost 15 us)
    for (int m = threadIdx.x; m < n_img; m += RANK_IT) {
        const int c = sel_count[m];
        s_cnt[m] = (c > 0 && c <= RANK_MAX) ? c : 0;
    }
    __syncthreads();
    int total = 0;
    for (int m = 0; m < n_img; ++m) {   // block-uniform
        const int c = s_cnt[m];
        total += ((c + RANK_IT - 1) / RANK_IT) * ((c + RANK_JT - 1) / RANK_JT);
    }
    for (int item = blockIdx.x; item < total; item += gridDim.x) {   // block-uniform trip count
        int img = 0, local = item, n_i = 0, nti = 1;
        for (; img < n_img; ++img) {
            n_i = s_cnt[img];
            nti = (n_i + RANK_IT - 1) / RANK_IT;
            const int cnt = nti * ((n_i + RANK_JT - 1) / RANK_JT);
            if (local < cnt) break;
            local -= cnt;
        }
        const int64_t base = (int64_t)img * cap_img;
        const int bj = local / nti, bi = local - bj * nti;
        const int i0 = bi * RANK_IT, j0 = bj * RANK_JT;
        const int jn = n_i - j0 < RANK_JT ? n_i - j0 : RANK_JT;
        __syncthreads();   // the previous pair's scan is over: the slice may be replaced
        for (int t = threadIdx.x; t < jn; t += RANK_IT) {
            const uint64_t g = ((uint64_t)(uint32_t)in_hi[base + j0 + t] << 32) | in_lo[base + j0 + t];   // ~score << 32 | cand
            keys[t][0] = g;
            keys[t][1] = p_key_of(g, label_bits);
        }
        __syncthreads();
        const int i = i0 + threadIdx.x;
        uint64_t gi = 0ull, pi = 0ull;   // threads past the end count nothing
        if (i < n_i) {
            gi = ((uint64_t)(uint32_t)in_hi[base + i] << 32) | in_lo[base + i];
            pi = p_key_of(gi, label_bits);
        }
        uint32_t cg = 0u, cp = 0u;
#pragma unroll 8
        for (int t = 0; t < jn; ++t) {   // wave-uniform LDS address: one broadcast 16-byte read per key
            const uint64_t kg = keys[t][0], kp = keys[t][1];
            cg += kg < gi ? 1u : 0u;
            cp += kp < pi ? 1u : 0u;
        }
        if (i < n_i) {
            atomicAdd(&rank_g[base + i], cg);
            atomicAdd(&rank_p[base + i], cp);
        }
    }
}

__global__ __launch_bounds__(RANK_IT) void scatter_ranks_kernel(const uint64_t* in_hi, const uint32_t* in_lo, const int* img_count, const int* sel_count, int cap_img,
                                                                int label_bits, const uint32_t* rank_g, const uint32_t* rank_p, uint64_t* ghi, uint32_t* glo,
                                                                uint64_t* phi, uint32_t* plo, uint8_t* keep, int* status) {
    __shared__ int s_off;
    const int img = blockIdx.y;
    const int n_i = sel_count[img];
    if (n_i > RANK_MAX) return;   // sort_image_kernel's image
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int raw = img_count[img];
        atomicAdd(&status[ST_NCAND], n_i);
        if (raw > cap_img) { atomicOr(&status[ST_OVERFLOW], YMI_STATUS_OVERFLOW_CAPACITY); atomicMax(&status[ST_RSV], raw); }
    }
    const int i = blockIdx.x * RANK_IT + threadIdx.x;
    if (blockIdx.x * RANK_IT >= n_i) return;
    if (threadIdx.x < 64) {   // first wave: offset of this image in the compact arrays = sum of the counts before it
        int part = 0;
        for (int j = threadIdx.x; j < img; j += 64) part += sel_count[j];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
        if (threadIdx.x == 0) s_off = part;
    }
    __syncthreads();
    if (i >= n_i) return;
    const int off = s_off;
    const int64_t base = (int64_t)img * cap_img;
    const uint64_t g = ((uint64_t)(uint32_t)in_hi[base + i] << 32) | in_lo[base + i];
    const uint32_t r = rank_g[base + i], p = rank_p[base + i];
    ghi[off + r] = ((uint64_t)(unsigned)img << 32) | (g >> 32);
    glo[off + r] = (uint32_t)g;
    keep[off + r] = 0;
    phi[off + p] = ((uint64_t)(unsigned)img << 16) | (g & ((1ull << label_bits) - 1ull));
    plo[off + p] = (uint32_t)off + r;
}

__global__ __launch_bounds__(1024) void sort_image_kernel(uint64_t* in_hi, uint32_t* in_lo, const int* img_count, const int* sel_count, int cap_img, int n_img,
                                                          int label_bits, int lds_keys, int rank_max, uint64_t* ghi, uint32_t* glo, uint64_t* phi, uint32_t* plo,
                                                          uint8_t* keep, int* status) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_key[];
    __shared__ int s_off;
    const int img = blockIdx.x;
    const int raw = img_count[img];
    const int n_i = sel_count[img];   // records that take part (== min(raw, cap_img) unless the prefix selection cut the image)
    if (n_i <= rank_max) return;      // sorted by rank_image_kernel / scatter_ranks_kernel
    if (threadIdx.x < 64) {   // first wave: offset of this image in the compact arrays = sum of the counts before it
        int part = 0;
        for (int j = threadIdx.x; j < img; j += 64) part += sel_count[j];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
        if (threadIdx.x == 0) s_off = part;
    }
    if (threadIdx.x == 0) {
        atomicAdd(&status[ST_NCAND], n_i);
        if (raw > cap_img) { atomicOr(&status[ST_OVERFLOW], YMI_STATUS_OVERFLOW_CAPACITY); atomicMax(&status[ST_RSV], raw); }
    }
    const int64_t base = (int64_t)img * cap_img;
    uint64_t* keys = in_hi + base;      // this image's region doubles as key scratch (u64 per record)
    uint32_t* rank = in_lo + base;      // ... and as rank scratch once the candidate ids moved into the keys
    const uint32_t lmask = (1u << label_bits) - 1u;
    if (n_i <= lds_keys) {
        // single run (the common case after the prefix selection): both sorts stay in LDS, HBM is touched only to read
        // the records once and to write G and P -- no scratch round trips between the phases
        __syncthreads();   // s_off
        const int off = s_off;
        int np2 = 64;
        while (np2 < n_i) np2 <<= 1;
        for (int i = threadIdx.x; i < np2; i += blockDim.x)
            lds_key[i] = i < n_i ? (((uint64_t)(uint32_t)keys[i] << 32) | rank[i]) : ~0ull;   // ~score << 32 | cand; padding sorts last
        __syncthreads();
        bitonic_sort_lds(lds_key, np2);   // G order: score descending, ties by candidate index
        for (int i = threadIdx.x; i < n_i; i += blockDim.x) {
            const uint64_t k = lds_key[i];
            ghi[off + i] = ((uint64_t)(unsigned)img << 32) | (k >> 32);
            glo[off + i] = (uint32_t)k;
            keep[off + i] = 0;
            lds_key[i] = ((uint64_t)((uint32_t)k & lmask) << 32) | (uint32_t)i;   // label << 32 | rank in G (same index read / written)
        }
        __syncthreads();
        bitonic_sort_lds(lds_key, np2);   // P order: by label, then by G rank
        for (int i = threadIdx.x; i < n_i; i += blockDim.x) {
            const uint64_t k = lds_key[i];
            phi[off + i] = ((uint64_t)(unsigned)img << 16) | (k >> 32);
            plo[off + i] = (uint32_t)off + (uint32_t)k;
        }
        return;
    }
    for (int i = threadIdx.x; i < n_i; i += blockDim.x)
        keys[i] = ((uint64_t)(uint32_t)keys[i] << 32) | rank[i];   // ~score << 32 | cand  (same index read / written)
    __syncthreads();
    const int off = s_off;
    // G order: score descending, ties by candidate index
    sort_runs(keys, n_i, lds_key, lds_keys, rank);
    for (int i = threadIdx.x; i < n_i; i += blockDim.x) {
        const uint64_t k = keys[i];
        const uint32_t r = rank[i];
        ghi[off + r] = ((uint64_t)(unsigned)img << 32) | (k >> 32);
        glo[off + r] = (uint32_t)k;
        keep[off + r] = 0;
        keys[i] = ((uint64_t)((uint32_t)k & lmask) << 32) | r;   // label << 32 | rank in G (unique)
    }
    __syncthreads();
    // P order: by label, then by G rank
    sort_runs(keys, n_i, lds_key, lds_keys, rank);
    for (int i = threadIdx.x; i < n_i; i += blockDim.x) {
        const uint64_t k = keys[i];
        const uint32_t r = rank[i];
        phi[off + r] = ((uint64_t)(unsigned)img << 16) | (k >> 32);
        plo[off + r] = (uint32_t)off + (uint32_t)k;
    }
}

// ------------------------------------------------------------------------------------------
// 3. class-aware NMS.  After the global sort G, records are re-keyed as
//      hi' = img << 16 | label,  lo' = r (rank in G)
//    and stably sorted by label, so each (label, img) segment lists its candidates in score order.
//    One wave per segment runs the exact greedy algorithm: 64 candidates per step are tested against
//    the kept boxes so far (uniform loads), then resolved among themselves with ballots/shuffles.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void make_label_records_kernel(const uint64_t* ghi, const uint32_t* glo, uint64_t* phi, uint32_t* plo,
                                                                 const int* status, int cap, int label_bits, uint8_t* keep) {
    const int n = ncand(status, cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned label = glo[i] & ((1u << label_bits) - 1u);
    const unsigned img = (unsigned)(ghi[i] >> 32);
    phi[i] = ((uint64_t)img << 16) | label;
    plo[i] = (uint32_t)i;
    keep[i] = 0;
}

__global__ __launch_bounds__(256) void find_segments_kernel(const uint64_t* phi, int* status, int cap, uint32_t* seg_start) {
    const int n = ncand(status, cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || phi[i] != phi[i - 1]) seg_start[atomicAdd(&status[ST_NSEG], 1)] = (uint32_t)i;
}

__device__ __forceinline__ bool iou_gt(const f32x4 bi, float area_i, const f32x4 bj, float area_j, float thr) {
    // torchvision nms: inter / (iarea + jarea - inter) > thr, fp32, no fused ops
    const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
    const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
    const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
    const float inter = __fmul_rn(w, h);
    const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_i, area_j), inter));
    return iou > thr;
}

constexpr int NMS_KCAP = 384;   // kept boxes per wave held in LDS (6 KiB / wave); the rest spills to HBM scratch

__global__ __launch_bounds__(256) void nms_segments_kernel(const uint64_t* phi, const uint32_t* plo, const uint64_t* ghi, const uint32_t* glo,
                                                           const float* boxes_all, int total_anchors, int label_bits, const int* status, int cap,
                                                           const uint32_t* seg_start, float* kept_box, uint8_t* keep, float thr) {
    __shared__ f32x4 kept_lds[4][NMS_KCAP];
    const int n = ncand(status, cap);
    const int nseg = status[ST_NSEG];
    const int lane = threadIdx.x & 63;
    const int wave_local = threadIdx.x >> 6;
    const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    f32x4* kl = kept_lds[wave_local];
    for (int s = wave_global; s < nseg; s += nwaves) {
        const int start = (int)seg_start[s];
        const uint64_t key = phi[start];
        int kept_cnt = 0;
        f32x4* kb = reinterpret_cast<f32x4*>(kept_box) + start;   // spill area: kept <= segment length
        for (int c = start;; c += 64) {
            const int p = c + lane;
            const bool valid = p < n && phi[p] == key;
            const uint64_t vmask = __ballot(valid);
            if (vmask == 0) break;
            f32x4 box = {0.f, 0.f, 0.f, 0.f};
            unsigned r = 0;
            if (valid) {
                r = plo[p];
                const unsigned img = (unsigned)(ghi[r] >> 32);
                const unsigned anchor = glo[r] >> label_bits;
                box = *reinterpret_cast<const f32x4*>(boxes_all + ((int64_t)img * total_anchors + anchor) * 4);
            }
            const float area = __fmul_rn(__fsub_rn(box[2], box[0]), __fsub_rn(box[3], box[1]));
            bool alive = valid;
            // against boxes already kept in this segment (higher score): LDS broadcast reads, HBM beyond NMS_KCAP
            const int in_lds = kept_cnt < NMS_KCAP ? kept_cnt : NMS_KCAP;
            for (int j = 0; j < in_lds; ++j) {
                const f32x4 k = kl[j];
                const float ka = __fmul_rn(__fsub_rn(k[2], k[0]), __fsub_rn(k[3], k[1]));
                if (alive && iou_gt(k, ka, box, area, thr)) alive = false;
            }
            for (int j = NMS_KCAP; j < kept_cnt; ++j) {
                const f32x4 k = kb[j];
                const float ka = __fmul_rn(__fsub_rn(k[2], k[0]), __fsub_rn(k[3], k[1]));
                if (alive && iou_gt(k, ka, box, area, thr)) alive = false;
            }
            // among the 64 candidates of this step, in order
            uint64_t mask = __ballot(alive);
            uint64_t todo = mask;
            while (todo) {
                const int i = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                f32x4 bi;
                bi[0] = __shfl(box[0], i, 64);
                bi[1] = __shfl(box[1], i, 64);
                bi[2] = __shfl(box[2], i, 64);
                bi[3] = __shfl(box[3], i, 64);
                const float ai = __shfl(area, i, 64);
                if (alive && lane > i && iou_gt(bi, ai, box, area, thr)) alive = false;
                mask = __ballot(alive);
                todo &= mask;
            }
            if (alive) {
                const int slot = kept_cnt + __popcll(mask & lt);
                if (slot < NMS_KCAP) kl[slot] = box; else kb[slot] = box;
                keep[r] = 1;
            }
            kept_cnt += __popcll(mask);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // kept boxes (LDS / HBM) visible to this wave's later reads
            if (vmask != ~0ull) break;  // last (partial) step of the segment
        }
    }
}

// ------------------------------------------------------------------------------------------
// 4. top-k + rescale.  One block per image: binary-search the image's range in G, then an ordered
//    compaction of the keep flags in rank order; first K kept are written to the fixed slab.
// ------------------------------------------------------------------------------------------
struct GatherArgs {
    const uint64_t* ghi;
    const uint32_t* glo;
    const uint8_t* keep;
    const float* boxes_all;
    const float* rescale;
    const int* status;
    const int* truncated;   // per image: 1 = only a score prefix of its candidates was processed (NULL: never)
    const int* img_count;   // per image: raw candidate count (per-image path; NULL: the global sort, whose status[ST_NCAND] is the raw count)
    float* out_slab;        // (n, 6K + 1) packed wire slab or NULL (ymi_post_desc.out_slab)
    int* status_rw;
    int cap, total_anchors, label_bits, K;
    float* out_boxes;
    float* out_scores;
    int64_t* out_labels;
    int* out_count;
};

__device__ int lower_bound_img(const uint64_t* ghi, int n, unsigned img) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((unsigned)(ghi[mid] >> 32) < img) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void gather_topk_kernel(const GatherArgs a) {
    __shared__ int wave_cnt[4];
    __shared__ int s_taken;
    __shared__ int s_last;
    const int n = ncand(a.status, a.cap);
    const unsigned img = blockIdx.x;
    const int begin = lower_bound_img(a.ghi, n, img), end = lower_bound_img(a.ghi, n, img + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (threadIdx.x == 0) s_taken = 0;
    __syncthreads();
    float gain = 0.f, padx = 0.f, pady = 0.f;
    if (a.rescale) { gain = a.rescale[img * 3]; padx = a.rescale[img * 3 + 1]; pady = a.rescale[img * 3 + 2]; }
    for (int c = begin; c < end; c += 256) {
        const int taken = s_taken;
        if (taken >= a.K) break;
        const int r = c + threadIdx.x;
        const bool k = r < end && a.keep[r] != 0;
        const uint64_t m = __ballot(k);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = taken;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        const int rank = off + __popcll(m & lt);
        if (k && rank < a.K) {
            const uint64_t hi = a.ghi[r];
            const uint32_t lo = a.glo[r];
            const unsigned anchor = lo >> a.label_bits;
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.boxes_all + ((int64_t)img * a.total_anchors + anchor) * 4);
            f32x4 o = b;
            if (gain > 0.f) {  // transform.py:362-365
                o[0] = __fdiv_rn(__fsub_rn(b[0], padx), gain);
                o[2] = __fdiv_rn(__fsub_rn(b[2], padx), gain);
                o[1] = __fdiv_rn(__fsub_rn(b[1], pady), gain);
                o[3] = __fdiv_rn(__fsub_rn(b[3], pady), gain);
            }
            const int64_t slot = (int64_t)img * a.K + rank;
            *reinterpret_cast<f32x4*>(a.out_boxes + slot * 4) = o;
            a.out_scores[slot] = __uint_as_float(~(uint32_t)hi);
            a.out_labels[slot] = (int64_t)(lo & ((1u << a.label_bits) - 1u));
            if (a.out_slab != nullptr) {   // the same detection in the packed wire slab (rows of 6K + 1 floats: only 4-byte aligned)
                float* row = a.out_slab + (int64_t)img * (6 * a.K + 1);
                row[4 * rank] = o[0]; row[4 * rank + 1] = o[1]; row[4 * rank + 2] = o[2]; row[4 * rank + 3] = o[3];
                row[4 * a.K + rank] = __uint_as_float(~(uint32_t)hi);
                row[5 * a.K + rank] = (float)(lo & ((1u << a.label_bits) - 1u));
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_taken = taken + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const int cnt = s_taken < a.K ? s_taken : a.K;
    if (a.out_slab != nullptr) {   // slots past the count are zero: the slab is a function of the detections alone
        float* row = a.out_slab + (int64_t)img * (6 * a.K + 1);
        for (int t = cnt + (int)threadIdx.x; t < a.K; t += 256) {
            row[4 * t] = 0.f; row[4 * t + 1] = 0.f; row[4 * t + 2] = 0.f; row[4 * t + 3] = 0.f;
            row[4 * a.K + t] = 0.f;
            row[5 * a.K + t] = 0.f;
        }
        if (threadIdx.x == 0) row[6 * a.K] = (float)cnt;
    }
    if (threadIdx.x == 0) {
        a.out_count[img] = cnt;
        // a truncated image must reach K survivors on its prefix alone, else the cut may have mattered
        if (a.truncated != nullptr && a.truncated[img] != 0 && s_taken < a.K) atomicOr(&a.status_rw[ST_OVERFLOW], YMI_STATUS_PREFIX_SHORT);
        if (a.img_count != nullptr) atomicAdd(&a.status_rw[ST_RAW], a.img_count[img]);
        else if (img == 0) atomicAdd(&a.status_rw[ST_RAW], n);
        // last-block fix-up: the status word is final only when every image's block has finished (ST_DONE is zeroed by post_begin)
        __threadfence();
        s_last = atomicAdd(&a.status_rw[ST_DONE], 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (s_last && a.out_slab != nullptr) {
        __threadfence();
        if (atomicOr(&a.status_rw[ST_OVERFLOW], 0) != 0)   // the caller will re-run this batch: mark every row stale (yolort_amd/dist.py SLAB_STALE)
            for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) a.out_slab[(int64_t)i * (6 * a.K + 1) + 6 * a.K] = YMI_SLAB_STALE;
    }
}

// ------------------------------------------------------------------------------------------
// host-side orchestration
// ------------------------------------------------------------------------------------------

struct SortState {
    int cur;  // index of the array pair holding the current order
};

static int radix_pass(const Workspace& w, SortState& st, int* status, int cap, int shift, hipStream_t s) {
    const int nblk = cdiv(cap, SORT_ITEMS);
    const int src = st.cur, dst = st.cur ^ 1;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(256), 0, s, w.hi[src], w.lo[src], status, cap, shift, w.hist);
    uint32_t* tot = w.hist + (int64_t)nblk * 256;  // 256 words after the per-block table
    hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(256), 0, s, status, cap, w.hist, tot);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(256), 0, s, w.hi[src], w.lo[src], w.hi[dst], w.lo[dst], status, cap, shift, w.hist, tot);
    st.cur = dst;
    return check_launch("radix pass");
}

// segments -> class-aware NMS -> top-k gather, given G (arrays w.hi/lo[g]) and the per-class order P
static int nms_gather(const Workspace& w, int g, const uint64_t* phi, const uint32_t* plo, int* status, int cap, int n_img, int label_bits, int total_anchors,
                      float nms_thresh, int K, const float* rescale, float* out_boxes, float* out_scores, int64_t* out_labels, int* out_count, hipStream_t s,
                      const int* truncated = nullptr, float* out_slab = nullptr, const int* img_count = nullptr) {
    const int nthr_blocks = cdiv(cap, 256);
    uint32_t* seg_start = w.seg_start;
    float* kept_box = w.kept_box;
    hipLaunchKernelGGL(find_segments_kernel, dim3(nthr_blocks), dim3(256), 0, s, phi, status, cap, seg_start);
    hipLaunchKernelGGL(nms_segments_kernel, dim3(1024), dim3(256), 0, s, phi, plo, w.hi[g], w.lo[g], w.boxes_all, total_anchors, label_bits, status, cap,
                       seg_start, kept_box, w.keep, nms_thresh);
    GatherArgs ga;
    ga.ghi = w.hi[g]; ga.glo = w.lo[g]; ga.keep = w.keep; ga.boxes_all = w.boxes_all; ga.rescale = rescale; ga.status = status;
    ga.truncated = truncated; ga.status_rw = status; ga.out_slab = out_slab; ga.img_count = img_count;
    ga.cap = cap; ga.total_anchors = total_anchors; ga.label_bits = label_bits; ga.K = K;
    ga.out_boxes = out_boxes; ga.out_scores = out_scores; ga.out_labels = out_labels; ga.out_count = out_count;
    hipLaunchKernelGGL(gather_topk_kernel, dim3(n_img), dim3(256), 0, s, ga);
    return check_launch("nms/gather");
}

// sorts records in w.hi/lo[st.cur] by (img, score desc, cand asc), then runs NMS + gather.
static int sort_nms_gather(const Workspace& w, SortState st, int* status, int cap, int n_img, int lo_bits, int label_bits, int total_anchors,
                           float nms_thresh, int K, const float* rescale, float* out_boxes, float* out_scores, int64_t* out_labels, int* out_count,
                           hipStream_t s, float* out_slab = nullptr) {
    int rc;
    for (int sh = 0; sh < lo_bits; sh += 8) if ((rc = radix_pass(w, st, status, cap, sh, s)) != YMI_OK) return rc;
    for (int sh = 32; sh < 64; sh += 8) if ((rc = radix_pass(w, st, status, cap, sh, s)) != YMI_OK) return rc;
    const int img_bits = bits_for(n_img);
    for (int sh = 64; sh < 64 + img_bits; sh += 8) if ((rc = radix_pass(w, st, status, cap, sh, s)) != YMI_OK) return rc;
    // G = current arrays; P = the other pair, re-keyed by label
    const int g = st.cur, p = st.cur ^ 1;
    const int nthr_blocks = cdiv(cap, 256);
    hipLaunchKernelGGL(make_label_records_kernel, dim3(nthr_blocks), dim3(256), 0, s, w.hi[g], w.lo[g], w.hi[p], w.lo[p], status, cap, label_bits, w.keep);
    // The label passes ping-pong between P and a scratch pair; G must stay intact, so P's partner
    // arrays are carved from seg_start/kept_box which are not live yet.
    Workspace wl = w;
    wl.hi[0] = w.hi[p];
    wl.lo[0] = w.lo[p];
    wl.hi[1] = reinterpret_cast<uint64_t*>(w.kept_box);        // cap * 16 bytes >= cap * 8
    wl.lo[1] = w.seg_start;                                   // cap * 4 bytes
    SortState sl{0};
    int label_passes = 0;
    for (int sh = 32; sh < 32 + label_bits; sh += 8) {
        if ((rc = radix_pass(wl, sl, status, cap, sh, s)) != YMI_OK) return rc;
        ++label_passes;
    }
    const uint64_t* phi = wl.hi[sl.cur];
    const uint32_t* plo = wl.lo[sl.cur];
    if (sl.cur == 1) {
        // sorted P currently lives in the scratch (kept_box/seg_start): move it back to P's own arrays
        YMI_CHECK_HIP(hipMemcpyAsync(w.hi[p], wl.hi[1], (size_t)cap * 8, hipMemcpyDeviceToDevice, s));
        YMI_CHECK_HIP(hipMemcpyAsync(w.lo[p], wl.lo[1], (size_t)cap * 4, hipMemcpyDeviceToDevice, s));
        phi = w.hi[p];
        plo = w.lo[p];
    }
    (void)label_passes;
    return nms_gather(w, g, phi, plo, status, cap, n_img, label_bits, total_anchors, nms_thresh, K, rescale, out_boxes, out_scores, out_labels, out_count, s, nullptr, out_slab);
}

static int post_validate(const ymi_post_desc* d, bool need_logits, PostLayout& L, Workspace& w) {
    YMI_REQUIRE(d != nullptr, "ymi_postprocess: null descriptor");
    YMI_REQUIRE(d->num_levels >= 1 && d->num_levels <= YMI_MAX_LEVELS, "ymi_postprocess: num_levels %d out of range", d->num_levels);
    YMI_REQUIRE(d->n >= 1 && d->n <= 65535, "ymi_postprocess: batch size %d out of range", d->n);
    YMI_REQUIRE(d->num_classes >= 1 && d->num_classes <= 4096, "ymi_postprocess: num_classes %d out of range", d->num_classes);
    YMI_REQUIRE(d->out_boxes && d->out_scores && d->out_labels && d->out_count && d->status && d->ws, "ymi_postprocess: null buffer");
    YMI_REQUIRE(d->detections_per_img >= 1 && d->cand_cap >= 1, "ymi_postprocess: detections_per_img and cand_cap must be positive");
    for (int l = 0; l < d->num_levels; ++l) {
        YMI_REQUIRE(d->lh[l] >= 1 && d->lw[l] >= 1, "ymi_postprocess: level %d has an empty grid", l);
        if (need_logits) {
            YMI_REQUIRE(d->logits[l] != nullptr, "ymi_postprocess: logits[%d] is null", l);
            YMI_REQUIRE(d->lcstride[l] >= 3 * (d->num_classes + 5) && d->lcstride[l] % 4 == 0, "ymi_postprocess: lcstride[%d]=%d too small / not a multiple of 4", l, d->lcstride[l]);
        }
    }
    L = post_layout(d);
    YMI_REQUIRE(L.label_bits + L.anchor_bits <= 32, "ymi_postprocess: %d anchors x %d classes exceed the 32-bit candidate index", L.total_anchors, d->num_classes);
    w = carve(d->ws, d->n, L.total_anchors, d->cand_cap);
    YMI_REQUIRE(d->ws_bytes >= w.total, "ymi_postprocess: workspace too small (%lld < %lld)", (long long)d->ws_bytes, (long long)w.total);
    return YMI_OK;
}

// the three counter arrays of a batch in ONE launch (three hipMemsetAsync were three fill kernels of ~5.7 us each on the conv stack's stream, in front of the head:
// profiles/r06u_layer_table_c2.csv "other kernels")
__global__ __launch_bounds__(256) void post_reset_kernel(int* a, int na, int* b, int nb, int* c, int nc) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < na + nb + nc; i += gridDim.x * 256) {
        if (i < na) a[i] = 0;
        else if (i < na + nb) b[i - na] = 0;
        else c[i - na - nb] = 0;
    }
}

// stage 1 of 3: reset the counters the candidate producers append to
int post_begin_launch(const ymi_post_desc* d, hipStream_t s) {
    PostLayout L;
    Workspace w;
    int rc = post_validate(d, false, L, w);
    if (rc != YMI_OK) return rc;
    const int nc = L.per_image ? d->n : 0;
    hipLaunchKernelGGL(post_reset_kernel, dim3(cdiv(ST_WORDS + d->n + nc, 256)), dim3(256), 0, s, (int*)d->status, (int)ST_WORDS, (int*)d->out_count, d->n, nc ? (int*)w.img_count : nullptr, nc);
    return check_launch("post_reset_kernel");
}

// stage 2 of 3 (unfused form): decode + threshold of the fp32 head outputs
static int post_decode_launch(const ymi_post_desc* d, hipStream_t s) {
    PostLayout L;
    Workspace w;
    int rc = post_validate(d, true, L, w);
    if (rc != YMI_OK) return rc;
    int level_off = 0;
    for (int l = 0; l < d->num_levels; ++l) {
        DecodeArgs a;
        a.logits = d->logits[l]; a.h = d->lh[l]; a.w = d->lw[l]; a.cs = d->lcstride[l]; a.stride = d->stride[l];
        for (int k = 0; k < 6; ++k) a.anc[k] = d->anchors[l][k];
        a.n = d->n; a.K = d->num_classes + 5; a.level_off = level_off;
        a.sink = make_sink(d, w, L);
        const int64_t npix = (int64_t)d->n * a.h * a.w;
        hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((npix + 4 * DEC_PIX_PER_WAVE - 1) / (4 * DEC_PIX_PER_WAVE))), dim3(256), 0, s, a);
        level_off += 3 * a.h * a.w;
    }
    return check_launch("decode_kernel");
}

// stage 3 of 3: sort, class-aware NMS, top-k + rescale, from the records the producers appended
int post_finish_launch(const ymi_post_desc* d, hipStream_t s) {
    PostLayout L;
    Workspace w;
    int rc = post_validate(d, false, L, w);
    if (rc != YMI_OK) return rc;
    if (L.per_image) {
        const int cap_img = L.cap_img;
        const bool exact_full = (d->flags & YMI_POST_EXACT_FULL) != 0;
        const int sel_t = exact_full ? 0 : (4 * d->detections_per_img > 4096 ? 4 * d->detections_per_img : 4096);
        // rank counters live in seg_start / kept_box, which are not in use before find_segments
        uint32_t* rank_g = w.seg_start;
        uint32_t* rank_p = reinterpret_cast<uint32_t*>(w.kept_box);
        // multi-block selection for large per-image regions; its scratch -- (n, 4096) bin and (n, 2048) digit histograms, n fill counters, n states -- lives in the P
        // arrays, which nothing uses before scatter_ranks_kernel (cap_img * 8 bytes per image against 24 KiB + 36)
        // (few images only: with 32 or more the one-block-per-image form already spreads over enough CUs, and the ladder's extra passes cost throughput)
        static const int multi_max_n = getenv("YOLORT_AMD_SEL_MULTI_MAXN") ? atoi(getenv("YOLORT_AMD_SEL_MULTI_MAXN")) : 32;   // tuning aid
        const bool multi = sel_t > 0 && cap_img >= 2 * SEL_SLICE && d->n < multi_max_n && getenv("YOLORT_AMD_SEL_SINGLE") == nullptr;
        int* ghist = reinterpret_cast<int*>(w.p_hi);
        int* rhist = ghist + (int64_t)d->n * SEL_BINS;
        int* gfill = rhist + (int64_t)d->n * SEL_RBINS;
        int* sel_state = gfill + d->n;
        const int nslices = cdiv(cap_img, SEL_SLICE);
        if (multi) {
            YMI_CHECK_HIP(hipMemsetAsync(ghist, 0, (size_t)d->n * (SEL_BINS + SEL_RBINS + 1 + SEL_STATE) * sizeof(int), s));
            hipLaunchKernelGGL(sel_hist_kernel, dim3(nslices, d->n), dim3(1024), 0, s, w.hi[0], w.img_count, cap_img, sel_t, ghist);
        }
        hipLaunchKernelGGL(select_prefix_kernel, dim3(d->n), dim3(1024), 0, s, w.hi[0], w.lo[0], w.img_count, cap_img, d->n, sel_t, w.sel_count, rank_g, rank_p,
                           multi ? ghist : nullptr, multi ? sel_state : nullptr);
        if (multi) {
            // the refinement ladder: a fixed number of (count, decide) pairs -- the launch sequence must not depend on the data; levels an image does not need return at once
            for (int l = 0; l < SEL_LEVELS; ++l) {
                hipLaunchKernelGGL(sel_rhist_kernel, dim3(nslices, d->n), dim3(1024), 0, s, w.hi[0], w.lo[0], w.img_count, cap_img, sel_state, rhist);
                hipLaunchKernelGGL(sel_rpick_kernel, dim3(d->n), dim3(64), 0, s, sel_state, rhist, w.sel_count);
            }
            hipLaunchKernelGGL(sel_compact_kernel, dim3(nslices, d->n), dim3(1024), 0, s, w.hi[0], w.lo[0], w.img_count, cap_img, sel_state, gfill, w.hi[1], w.lo[1]);
            hipLaunchKernelGGL(sel_copyback_kernel, dim3(cdiv(RANK_MAX, 256), d->n), dim3(256), 0, s, w.hi[0], w.lo[0], w.hi[1], w.lo[1], sel_state, w.sel_count, cap_img);
        }
        const int rank_cap = cap_img < RANK_MAX ? cap_img : RANK_MAX;   // no image holds more than cap_img records
        YMI_REQUIRE(d->n <= RANK_NIMG, "ymi_postprocess: more than %d images in a batch", RANK_NIMG);
        hipLaunchKernelGGL(rank_image_kernel, dim3(RANK_GRID), dim3(RANK_IT), 0, s, w.hi[0], w.lo[0], w.sel_count, d->n, cap_img,
                           L.label_bits, rank_g, rank_p);
        hipLaunchKernelGGL(scatter_ranks_kernel, dim3(cdiv(rank_cap, RANK_IT), d->n), dim3(RANK_IT), 0, s, w.hi[0], w.lo[0], w.img_count, w.sel_count, cap_img, L.label_bits,
                           rank_g, rank_p, w.hi[1], w.lo[1], w.p_hi, w.p_lo, w.keep, d->status);
        // LDS run length of the sort: with the prefix selection images rarely exceed 8192 records (longer ones take the
        // multi-run merge path), and 64 KiB instead of 128 leaves room for a convolution block on the same CU
        const int run_max = exact_full ? IMG_SORT_MAX : IMG_SORT_MAX / 2;
        const int lds_keys = cap_img < run_max ? cap_img : run_max;
        const size_t lds = (size_t)lds_keys * 8;
        if (lds > 64 * 1024) { const int rc_lds = allow_big_lds((const void*)sort_image_kernel, (int)lds); if (rc_lds != YMI_OK) return rc_lds; }
        // the producers wrote arrays [0] (per-image regions); G goes to arrays [1] (compact), P to its own pair
        if (cap_img > RANK_MAX)   // only then can an image exceed the ranking path
            hipLaunchKernelGGL(sort_image_kernel, dim3(d->n), dim3(1024), lds, s, w.hi[0], w.lo[0], w.img_count, w.sel_count, cap_img, d->n, L.label_bits, lds_keys,
                               RANK_MAX, w.hi[1], w.lo[1], w.p_hi, w.p_lo, w.keep, d->status);
        if ((rc = check_launch("select_prefix/sort_image")) != YMI_OK) return rc;
        return nms_gather(w, 1, w.p_hi, w.p_lo, d->status, d->cand_cap, d->n, L.label_bits, L.total_anchors, d->nms_thresh, d->detections_per_img, d->rescale,
                          d->out_boxes, d->out_scores, d->out_labels, d->out_count, s, w.sel_count + d->n, d->out_slab, w.img_count);
    }
    hipLaunchKernelGGL(finalize_count_kernel, dim3(1), dim3(64), 0, s, d->status, d->cand_cap);
    rc = check_launch("finalize_count");
    if (rc != YMI_OK) return rc;
    return sort_nms_gather(w, SortState{0}, d->status, d->cand_cap, d->n, L.label_bits + L.anchor_bits, L.label_bits, L.total_anchors, d->nms_thresh,
                           d->detections_per_img, d->rescale, d->out_boxes, d->out_scores, d->out_labels, d->out_count, s, d->out_slab);
}

int postprocess_launch(const ymi_post_desc* d, hipStream_t s) {
    int rc = post_begin_launch(d, s);
    if (rc != YMI_OK) return rc;
    if ((rc = post_decode_launch(d, s)) != YMI_OK) return rc;
    return post_finish_launch(d, s);
}

// ---- stand-alone batched NMS (one image) -------------------------------------------------
constexpr int NMS_LABEL_BITS = 12;

__global__ __launch_bounds__(256) void nms_make_records_kernel(const float* boxes, const float* scores, const int* labels, int n, float* boxes_all,
                                                               uint64_t* hi, uint32_t* lo, int* status) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { status[ST_NCAND] = n; status[ST_OVERFLOW] = 0; status[ST_NSEG] = 0; status[ST_RSV] = 0; }
    if (i >= n) return;
    *reinterpret_cast<f32x4*>(boxes_all + (int64_t)i * 4) = *reinterpret_cast<const f32x4*>(boxes + (int64_t)i * 4);
    hi[i] = (uint64_t)(~__float_as_uint(scores[i]));
    lo[i] = ((unsigned)i << NMS_LABEL_BITS) | ((unsigned)labels[i] & ((1u << NMS_LABEL_BITS) - 1u));
}

__global__ __launch_bounds__(256) void nms_emit_keep_kernel(const uint32_t* glo, const uint8_t* keep, int n, int* keep_out, int* count_out) {
    // single block: ordered compaction of kept ranks -> original indices
    __shared__ int wave_cnt[4];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int c = 0; c < n; c += 256) {
        const int r = c + threadIdx.x;
        const bool k = r < n && keep[r] != 0;
        const uint64_t m = __ballot(k);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (k) keep_out[off + __popcll(m & lt)] = (int)(glo[r] >> NMS_LABEL_BITS);
        __syncthreads();
        if (threadIdx.x == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) *count_out = s_base;
}

}  // namespace ymi

using namespace ymi;

extern "C" int64_t ymi_postprocess_ws_bytes(int n, int total_anchors, int cand_cap) {
    if (n <= 0 || total_anchors <= 0 || cand_cap <= 0) return 0;
    return carve(nullptr, n, total_anchors, cand_cap).total;
}

extern "C" int ymi_postprocess(const ymi_post_desc* d, void* stream) { return postprocess_launch(d, (hipStream_t)stream); }
extern "C" int ymi_post_begin(const ymi_post_desc* d, void* stream) { return post_begin_launch(d, (hipStream_t)stream); }
extern "C" int ymi_post_finish(const ymi_post_desc* d, void* stream) { return post_finish_launch(d, (hipStream_t)stream); }

extern "C" int64_t ymi_nms_ws_bytes(int n) {
    if (n <= 0) n = 1;
    return carve(nullptr, 1, n, n).total + 256 /* status */ + 2 * 256;
}

extern "C" int ymi_batched_nms(const float* boxes, const float* scores, const int32_t* labels, int n, float nms_thresh, int32_t* keep_out,
                               int32_t* count_out, void* ws, int64_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    YMI_REQUIRE(count_out && ws, "ymi_batched_nms: null buffer");
    if (n == 0) {
        YMI_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int), s));
        return YMI_OK;
    }
    YMI_REQUIRE(boxes && scores && labels && keep_out, "ymi_batched_nms: null buffer");
    YMI_REQUIRE(n > 0 && n < (1 << (32 - NMS_LABEL_BITS)), "ymi_batched_nms: n=%d out of range (max %d)", n, (1 << (32 - NMS_LABEL_BITS)) - 1);
    YMI_REQUIRE(ws_bytes >= ymi_nms_ws_bytes(n), "ymi_batched_nms: workspace too small");
    int* status = (int*)ws;
    const Workspace w = carve((char*)ws + 256, 1, n, n);
    hipLaunchKernelGGL(nms_make_records_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, boxes, scores, labels, n, w.boxes_all, w.hi[0], w.lo[0], status);
    // dummy slab outputs for the shared pipeline: carve them after the workspace proper is not possible
    // without more memory, so the stand-alone path skips gather and emits kept indices itself.
    int rc;
    SortState st{0};
    const int lo_bits = NMS_LABEL_BITS + bits_for(n);
    for (int sh = 0; sh < lo_bits; sh += 8) if ((rc = radix_pass(w, st, status, n, sh, s)) != YMI_OK) return rc;
    for (int sh = 32; sh < 64; sh += 8) if ((rc = radix_pass(w, st, status, n, sh, s)) != YMI_OK) return rc;
    const int g = st.cur, p = st.cur ^ 1;
    hipLaunchKernelGGL(make_label_records_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, w.hi[g], w.lo[g], w.hi[p], w.lo[p], status, n, NMS_LABEL_BITS, w.keep);
    Workspace wl = w;
    wl.hi[0] = w.hi[p]; wl.lo[0] = w.lo[p];
    wl.hi[1] = reinterpret_cast<uint64_t*>(w.kept_box); wl.lo[1] = w.seg_start;
    SortState sl{0};
    for (int sh = 32; sh < 32 + NMS_LABEL_BITS; sh += 8) if ((rc = radix_pass(wl, sl, status, n, sh, s)) != YMI_OK) return rc;
    const uint64_t* phi = wl.hi[sl.cur];
    const uint32_t* plo = wl.lo[sl.cur];
    if (sl.cur == 1) {
        YMI_CHECK_HIP(hipMemcpyAsync(w.hi[p], wl.hi[1], (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        YMI_CHECK_HIP(hipMemcpyAsync(w.lo[p], wl.lo[1], (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        phi = w.hi[p]; plo = w.lo[p];
    }
    hipLaunchKernelGGL(find_segments_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, phi, status, n, w.seg_start);
    hipLaunchKernelGGL(nms_segments_kernel, dim3(256), dim3(256), 0, s, phi, plo, w.hi[g], w.lo[g], w.boxes_all, n, NMS_LABEL_BITS, status, n, w.seg_start,
                       w.kept_box, w.keep, nms_thresh);
    hipLaunchKernelGGL(nms_emit_keep_kernel, dim3(1), dim3(256), 0, s, w.lo[g], w.keep, n, keep_out, count_out);
    return check_launch("ymi_batched_nms");
}
