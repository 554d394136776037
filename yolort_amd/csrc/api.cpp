// C-ABI glue of libyolort_amd.so: error reporting and the plan executor (recorded launch sequence,
// optional hipGraph replay, per-op HIP-event profiling).  See include/yolort_amd.h.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <map>
#include <utility>
#include <vector>

#include "common.hpp"

namespace ymi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int allow_big_lds(const void* kernel_fn, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> done;   // (kernel, device) -> largest size opted in so far
    int dev = 0;
    YMI_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find({kernel_fn, dev});
    if (it != done.end() && it->second >= bytes) return YMI_OK;
    YMI_CHECK_HIP(hipFuncSetAttribute(kernel_fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[{kernel_fn, dev}] = bytes;
    return YMI_OK;
}

size_t lds_floor_bytes() {
    static const size_t v = [] {
        const char* e = getenv("YOLORT_AMD_LDS_FLOOR_KB");
        return e ? (size_t)atoi(e) * 1024 : (size_t)0;
    }();
    return v;
}

bool head_anchor_split() {
    static const bool v = [] {
        const char* e = getenv("YOLORT_AMD_HEAD_SPLIT");
        return e ? atoi(e) != 0 : true;
    }();
    return v;
}

int conv2d_launch(const ymi_conv_desc* d, hipStream_t s);
int c3_fused_launch(const ymi_c3_desc* d, hipStream_t s);
int stem_body1_desc_launch(const ymi_conv_desc* stem, const ymi_conv_desc* body1, hipStream_t s);
int postprocess_launch(const ymi_post_desc* d, hipStream_t s);
int post_begin_launch(const ymi_post_desc* d, hipStream_t s);
int post_finish_launch(const ymi_post_desc* d, hipStream_t s);
int conv_head_decode_launch(const ymi_conv_desc* d, const ymi_post_desc* post, int level, hipStream_t s);
int conv_head_decode_group_launch(const ymi_conv_desc* descs, int n_levels, const ymi_post_desc* post, hipStream_t s);

enum OpKind { OP_CONV, OP_SPP, OP_UP, OP_COPY, OP_POST, OP_POST_BEGIN, OP_HEAD_DECODE, OP_HEAD_GROUP, OP_POST_FINISH, OP_C3_FUSED, OP_ACT };

struct Op {
    OpKind kind;
    ymi_conv_desc conv;
    ymi_conv_desc convs[YMI_MAX_LEVELS];   // OP_HEAD_GROUP: one head per pyramid level
    ymi_post_desc post;
    ymi_c3_desc c3;
    // generic small-op arguments
    const void* x;
    void* y;
    int i[8];
};

static int run_op(const Op& op, hipStream_t s) {
    switch (op.kind) {
        case OP_CONV: return conv2d_launch(&op.conv, s);
        case OP_SPP: return ymi_spp_pool(op.y, op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], s);
        case OP_UP: return ymi_upsample2x(op.x, op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.y, op.i[5], op.i[6], s);
        case OP_COPY: return ymi_copy_view(op.x, op.i[0], op.i[1], op.i[2], op.y, op.i[3], op.i[4], s);
        case OP_POST: return postprocess_launch(&op.post, s);
        case OP_POST_BEGIN: return post_begin_launch(&op.post, s);
        case OP_HEAD_DECODE: return conv_head_decode_launch(&op.conv, &op.post, op.i[0], s);
        case OP_HEAD_GROUP: return conv_head_decode_group_launch(op.convs, op.i[0], &op.post, s);
        case OP_POST_FINISH: return post_finish_launch(&op.post, s);
        case OP_C3_FUSED: return c3_fused_launch(&op.c3, s);
        case OP_ACT: return ymi_act(op.y, op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.x, op.i[5], s);
    }
    set_error("unknown op kind");
    return YMI_EINVAL;
}

}  // namespace ymi

struct ymi_plan {
    std::vector<ymi::Op> ops;
    bool fuse_stem = false;   // ops 0 + 1 (stem, body.1) run as ONE launch whenever a run covers both (ymi_plan_set_fuse_stem)
    // captured op ranges: the conv stack and the post-process of a batch are one graph launch each (ymi_plan_submit); a third range evicts the older of the two
    struct Captured {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        int first = -1, last = -1;
    } captured[2];
    int captured_next = 0;
    // ymi_plan_begin / ymi_plan_submit: inputs ready (caller's stream), conv stack done (main stream), batch done (side stream); created at first use
    hipEvent_t ev_in = nullptr, ev_conv = nullptr, ev_done = nullptr;
    bool submitted = false;
    std::vector<void*> owned;   // device memory of an imported plan (ymi_plan_import): freed with it
};

using namespace ymi;

extern "C" int ymi_abi_version(void) { return YMI_ABI_VERSION; }
extern "C" const char* ymi_last_error(void) { return g_err; }
extern "C" int ymi_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return YMI_EHIP;
    }
    return n;
}

// ---- measurement aid: the shader clock while the chip is busy (s_memtime counts shader cycles, s_memrealtime a constant 100 MHz) ----
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = t1 - t0;
    out[1] = r1 - r0;
}
extern "C" int ymi_clock_probe(uint64_t* out, int spin_us, void* stream) {
    YMI_REQUIRE(out != nullptr && spin_us >= 1 && spin_us <= 100000, "ymi_clock_probe: out must be device memory, 1 <= spin_us <= 100000");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, (unsigned long long)spin_us * 100ull);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("ymi_clock_probe: launch failed: %s", hipGetErrorString(e));
        return YMI_EHIP;
    }
    return YMI_OK;
}

extern "C" ymi_plan* ymi_plan_create(void) { return new (std::nothrow) ymi_plan(); }

static void drop_captured(ymi_plan::Captured& c) {
    if (c.exec) (void)hipGraphExecDestroy(c.exec);
    if (c.graph) (void)hipGraphDestroy(c.graph);
    c = ymi_plan::Captured();
}
static void drop_graph(ymi_plan* p) {
    for (auto& c : p->captured) drop_captured(c);
}

extern "C" void ymi_plan_destroy(ymi_plan* p) {
    if (!p) return;
    drop_graph(p);
    for (hipEvent_t e : {p->ev_in, p->ev_conv, p->ev_done})
        if (e) (void)hipEventDestroy(e);
    for (void* m : p->owned) (void)hipFree(m);
    delete p;
}

extern "C" int ymi_plan_add_conv(ymi_plan* p, const ymi_conv_desc* d) {
    YMI_REQUIRE(p && d, "ymi_plan_add_conv: null argument");
    Op op;
    memset(&op, 0, sizeof(op));
    op.kind = OP_CONV;
    op.conv = *d;
    p->ops.push_back(op);
    drop_graph(p);
    return (int)p->ops.size() - 1;
}

extern "C" int ymi_plan_add_c3_fused(ymi_plan* p, const ymi_c3_desc* d) {
    YMI_REQUIRE(p && d, "ymi_plan_add_c3_fused: null argument");
    Op op;
    memset(&op, 0, sizeof(op));
    op.kind = OP_C3_FUSED;
    op.c3 = *d;
    p->ops.push_back(op);
    drop_graph(p);
    return (int)p->ops.size() - 1;
}

extern "C" int ymi_plan_add_spp_pool(ymi_plan* p, void* buf, int n, int h, int w, int c, int cstride, int dtype) {
    YMI_REQUIRE(p && buf, "ymi_plan_add_spp_pool: null argument");
    Op op;
    memset(&op, 0, sizeof(op));
    op.kind = OP_SPP;
    op.y = buf;
    op.i[0] = n; op.i[1] = h; op.i[2] = w; op.i[3] = c; op.i[4] = cstride; op.i[5] = dtype;
    p->ops.push_back(op);
    drop_graph(p);
    return (int)p->ops.size() - 1;
}

extern "C" int ymi_plan_add_upsample2x(ymi_plan* p, const void* x, int x_cstride, int n, int h, int w, int c, void* y, int y_cstride, int dtype) {
    YMI_REQUIRE(p && x && y, "ymi_plan_add_upsample2x: null argument");
    Op op;
    memset(&op, 0, sizeof(op));
    op.kind = OP_UP;
    op.x = x; op.y = y;
    op.i[0] = x_cstride; op.i[1] = n; op.i[2] = h; op.i[3] = w; op.i[4] = c; op.i[5] = y_cstride; op.i[6] = dtype;
    p->ops.push_back(op);
    drop_graph(p);
    return (int)p->ops.size() - 1;
}

extern "C" int ymi_plan_add_copy_view(ymi_plan* p, const void* x, int x_cstride, int npix, int c, void* y, int y_cstride, int dtype) {
    YMI_REQUIRE(p && x && y, "ymi_plan_add_copy_view: null argument");
    Op op;
    memset(&op, 0, sizeof(op));
    op.kind = OP_COPY;
    op.x = x; op.y = y;
    op.i[0] = x_cstride; op.i[1] = npix; op.i[2] = c; op.i[3] = y_cstride; op.i[4] = dtype;
    p->ops.push_back(op);
    drop_graph(p);
    return (int)p->ops.size() - 1;
}

extern "C" int ymi_plan_add_act(ymi_plan* p, void* y, int y_cstride, int npix, int c, int dtype, int act, const void* res, int res_cstride) {
    YMI_REQUIRE(p && y, "ymi_plan_add_act: null argument");
    Op op;
    memset(&op, 0, sizeof(op));
    op.kind = OP_ACT;
    op.x = res; op.y = y;
    op.i[0] = y_cstride; op.i[1] = npix; op.i[2] = c; op.i[3] = dtype; op.i[4] = act; op.i[5] = res_cstride;
    p->ops.push_back(op);
    drop_graph(p);
    return (int)p->ops.size() - 1;
}

extern "C" int ymi_plan_add_postprocess(ymi_plan* p, const ymi_post_desc* d) {
    YMI_REQUIRE(p && d, "ymi_plan_add_postprocess: null argument");
    Op op;
    memset(&op, 0, sizeof(op));
    op.kind = OP_POST;
    op.post = *d;
    p->ops.push_back(op);
    drop_graph(p);
    return (int)p->ops.size() - 1;
}

static int add_post_op(ymi_plan* p, OpKind kind, const ymi_post_desc* d, const ymi_conv_desc* conv, int level) {
    Op op;
    memset(&op, 0, sizeof(op));
    op.kind = kind;
    op.post = *d;
    if (conv) op.conv = *conv;
    op.i[0] = level;
    p->ops.push_back(op);
    drop_graph(p);
    return (int)p->ops.size() - 1;
}

extern "C" int ymi_plan_add_post_begin(ymi_plan* p, const ymi_post_desc* d) {
    YMI_REQUIRE(p && d, "ymi_plan_add_post_begin: null argument");
    return add_post_op(p, OP_POST_BEGIN, d, nullptr, 0);
}

extern "C" int ymi_plan_add_head_decode(ymi_plan* p, const ymi_conv_desc* conv, const ymi_post_desc* d, int level) {
    YMI_REQUIRE(p && conv && d, "ymi_plan_add_head_decode: null argument");
    return add_post_op(p, OP_HEAD_DECODE, d, conv, level);
}

extern "C" int ymi_plan_add_head_decode_group(ymi_plan* p, const ymi_conv_desc* convs, int n_levels, const ymi_post_desc* d) {
    YMI_REQUIRE(p && convs && d && n_levels >= 1 && n_levels <= YMI_MAX_LEVELS, "ymi_plan_add_head_decode_group: bad argument");
    Op op;
    memset(&op, 0, sizeof(op));
    op.kind = OP_HEAD_GROUP;
    op.post = *d;
    for (int l = 0; l < n_levels; ++l) op.convs[l] = convs[l];
    op.i[0] = n_levels;
    p->ops.push_back(op);
    drop_graph(p);
    return (int)p->ops.size() - 1;
}

extern "C" int ymi_plan_add_post_finish(ymi_plan* p, const ymi_post_desc* d) {
    YMI_REQUIRE(p && d, "ymi_plan_add_post_finish: null argument");
    return add_post_op(p, OP_POST_FINISH, d, nullptr, 0);
}

extern "C" int ymi_plan_num_ops(const ymi_plan* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int ymi_plan_set_fuse_stem(ymi_plan* p, int on) {
    YMI_REQUIRE(p != nullptr, "ymi_plan_set_fuse_stem: null plan");
    if (on) {
        YMI_REQUIRE(p->ops.size() >= 2 && p->ops[0].kind == OP_CONV && p->ops[1].kind == OP_CONV, "ymi_plan_set_fuse_stem: ops 0 and 1 must be convolutions");
        YMI_REQUIRE(p->ops[1].conv.x == p->ops[0].conv.y && p->ops[1].conv.x_cstride == p->ops[0].conv.y_cstride, "ymi_plan_set_fuse_stem: op 1 must read op 0's output");
    }
    if (p->fuse_stem != (on != 0)) drop_graph(p);
    p->fuse_stem = on != 0;
    return YMI_OK;
}

// ops [first, last) in order; with fuse_stem, ops 0 + 1 together when the range holds both (i advances past op 1)
static int run_range_op(const ymi_plan* p, int& i, int last, hipStream_t s) {
    if (i == 0 && p->fuse_stem && last >= 2) {
        i = 1;
        return stem_body1_desc_launch(&p->ops[0].conv, &p->ops[1].conv, s);
    }
    return run_op(p->ops[i], s);
}

extern "C" int ymi_plan_run(ymi_plan* p, int first, int last, int use_graph, void* stream) {
    YMI_REQUIRE(p != nullptr, "ymi_plan_run: null plan");
    hipStream_t s = (hipStream_t)stream;
    const int nops = (int)p->ops.size();
    if (last < 0 || last > nops) last = nops;
    if (first < 0) first = 0;
    if (!use_graph) {
        for (int i = first; i < last; ++i) {
            int rc = run_range_op(p, i, last, s);
            if (rc != YMI_OK) return rc;
        }
        return YMI_OK;
    }
    ymi_plan::Captured* c = nullptr;
    for (auto& k : p->captured)
        if (k.exec && k.first == first && k.last == last) c = &k;
    if (c == nullptr) {
        for (auto& k : p->captured)
            if (!k.exec && c == nullptr) c = &k;
        if (c == nullptr) c = &p->captured[p->captured_next];
        p->captured_next = (int)(c - p->captured) ^ 1;
        drop_captured(*c);
        YMI_REQUIRE(s != nullptr, "ymi_plan_run: graph capture needs a non-default stream");
        YMI_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = YMI_OK;
        for (int i = first; i < last && rc == YMI_OK; ++i) rc = run_range_op(p, i, last, s);
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(s, &g);
        if (rc != YMI_OK) {
            if (g) (void)hipGraphDestroy(g);
            return rc;
        }
        if (e != hipSuccess) {
            set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
            return YMI_EHIP;
        }
        c->graph = g;
        YMI_CHECK_HIP(hipGraphInstantiate(&c->exec, c->graph, nullptr, nullptr, 0));
        c->first = first;
        c->last = last;
    }
    YMI_CHECK_HIP(hipGraphLaunch(c->exec, s));
    return YMI_OK;
}

static int plan_events(ymi_plan* p) {
    if (p->ev_done) return YMI_OK;
    YMI_CHECK_HIP(hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming));
    YMI_CHECK_HIP(hipEventCreateWithFlags(&p->ev_conv, hipEventDisableTiming));
    YMI_CHECK_HIP(hipEventCreateWithFlags(&p->ev_done, hipEventDisableTiming));
    return YMI_OK;
}

extern "C" int ymi_plan_begin(ymi_plan* p, void* caller_stream, void* main_stream) {
    YMI_REQUIRE(p != nullptr, "ymi_plan_begin: null plan");
    int rc = plan_events(p);
    if (rc != YMI_OK) return rc;
    hipStream_t cs = (hipStream_t)caller_stream, ms = (hipStream_t)main_stream;
    if (cs != ms) {
        YMI_CHECK_HIP(hipEventRecord(p->ev_in, cs));
        YMI_CHECK_HIP(hipStreamWaitEvent(ms, p->ev_in, 0));
    }
    if (p->submitted) YMI_CHECK_HIP(hipStreamWaitEvent(ms, p->ev_done, 0));
    return YMI_OK;
}

extern "C" int ymi_plan_submit(ymi_plan* p, int first, int n_conv, int use_graph, void* main_stream, void* side_stream, const void* dev_result, void* host_result,
                               size_t result_bytes, int main_waits_done) {
    YMI_REQUIRE(p != nullptr, "ymi_plan_submit: null plan");
    const int nops = (int)p->ops.size();
    YMI_REQUIRE(first >= 0 && first <= n_conv && n_conv <= nops, "ymi_plan_submit: need 0 <= first <= n_conv <= num_ops");
    YMI_REQUIRE(result_bytes == 0 || (dev_result != nullptr && host_result != nullptr), "ymi_plan_submit: result_bytes without buffers");
    int rc = plan_events(p);
    if (rc != YMI_OK) return rc;
    hipStream_t ms = (hipStream_t)main_stream, ss = (hipStream_t)side_stream;
    if (first < n_conv) {
        rc = ymi_plan_run(p, first, n_conv, use_graph & 1, main_stream);
        if (rc != YMI_OK) return rc;
    }
    if (ss != ms) {
        YMI_CHECK_HIP(hipEventRecord(p->ev_conv, ms));
        YMI_CHECK_HIP(hipStreamWaitEvent(ss, p->ev_conv, 0));
    }
    if (n_conv < nops) {
        rc = ymi_plan_run(p, n_conv, nops, (use_graph & 2) && ss != nullptr ? 1 : 0, side_stream);   // bit 1: the post-process range as a graph launch of its own
        if (rc != YMI_OK) return rc;
    }
    if (result_bytes) YMI_CHECK_HIP(hipMemcpyAsync(host_result, dev_result, result_bytes, hipMemcpyDeviceToHost, ss));
    YMI_CHECK_HIP(hipEventRecord(p->ev_done, ss));
    p->submitted = true;
    if (main_waits_done && ss != ms) YMI_CHECK_HIP(hipStreamWaitEvent(ms, p->ev_done, 0));
    return YMI_OK;
}

extern "C" int ymi_plan_done_query(ymi_plan* p) {
    YMI_REQUIRE(p != nullptr, "ymi_plan_done_query: null plan");
    if (!p->submitted) return 1;
    const hipError_t e = hipEventQuery(p->ev_done);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();   // not an error: clear the sticky code
        return 0;
    }
    set_error("ymi_plan_done_query: hipEventQuery failed: %s", hipGetErrorString(e));
    return YMI_EHIP;
}

extern "C" int ymi_plan_done_sync(ymi_plan* p) {
    YMI_REQUIRE(p != nullptr, "ymi_plan_done_sync: null plan");
    if (p->submitted) YMI_CHECK_HIP(hipEventSynchronize(p->ev_done));
    return YMI_OK;
}

extern "C" int ymi_plan_profile(ymi_plan* p, int iters, float* ms_out, void* stream) {
    YMI_REQUIRE(p && ms_out && iters > 0, "ymi_plan_profile: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int nops = (int)p->ops.size();
    std::vector<hipEvent_t> ev(nops + 1);
    for (auto& e : ev) YMI_CHECK_HIP(hipEventCreate(&e));
    for (int i = 0; i < nops; ++i) ms_out[i] = 0.f;
    int rc = YMI_OK;
    for (int it = 0; it < iters && rc == YMI_OK; ++it) {
        YMI_CHECK_HIP(hipEventRecord(ev[0], s));
        for (int i = 0; i < nops && rc == YMI_OK; ++i) {
            const int i0 = i;
            rc = run_range_op(p, i, nops, s);   // a fused stem: its time lands on op 0, op 1 reads 0
            for (int k = i0; k <= i; ++k) YMI_CHECK_HIP(hipEventRecord(ev[k + 1], s));
        }
        YMI_CHECK_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < nops && rc == YMI_OK; ++i) {
            float ms = 0.f;
            YMI_CHECK_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            ms_out[i] += ms / (float)iters;
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}


// ---------------------------------------------------------------------------------------------------
// Plan export / import (include/yolort_amd.h): a recorded plan as a self-contained file -- the launch descriptors with every pointer rewritten as
// (memory region, offset), the regions' sizes and roles, and the contents of the constant ones (packed weights, biases, im2col tables, weight streams).
// A consumer without Python (a C++ server; the role yolort/runtime's TorchScript / ONNX artefacts play for the reference) calls ymi_plan_import, copies its
// letterboxed batch into the INPUT region, runs ymi_plan_run and reads the OUTPUT regions.
// ---------------------------------------------------------------------------------------------------
namespace ymi {
template <class F>
static void for_each_pointer(Op& op, F&& f) {   // every device-pointer field of an op (unused ones are NULL: ops are zero-initialised)
    auto conv = [&](ymi_conv_desc& c) {
        f((void**)&c.x); f((void**)&c.w); f((void**)&c.bias); f((void**)&c.ktab); f((void**)&c.y); f((void**)&c.res); f((void**)&c.y2);
        f((void**)&c.chain_w); f((void**)&c.chain_bias); f((void**)&c.chain_y); f((void**)&c.chain_x2); f((void**)&c.zeros);
    };
    conv(op.conv);
    for (int l = 0; l < YMI_MAX_LEVELS; ++l) conv(op.convs[l]);
    ymi_post_desc& q = op.post;
    for (int l = 0; l < YMI_MAX_LEVELS; ++l) f((void**)&q.logits[l]);
    f((void**)&q.rescale); f((void**)&q.out_boxes); f((void**)&q.out_scores); f((void**)&q.out_labels); f((void**)&q.out_count); f((void**)&q.status);
    f((void**)&q.ws); f((void**)&q.out_slab);
    ymi_c3_desc& c3 = op.c3;
    f((void**)&c3.x); f((void**)&c3.y); f((void**)&c3.w12); f((void**)&c3.b12); f((void**)&c3.wm1); f((void**)&c3.bm1); f((void**)&c3.wm2); f((void**)&c3.bm2);
    f((void**)&c3.w3); f((void**)&c3.b3); f((void**)&c3.wblob); f((void**)&c3.y1_in); f((void**)&c3.y1_out); f((void**)&c3.y2);
    f((void**)&op.x); f((void**)&op.y);
}
struct PlanFileHeader {
    char magic[8];            // "YMIPLAN1"
    int32_t abi, n_regions, n_ops, fuse_stem;
    int64_t op_bytes;         // sizeof(Op) of the writer: a reader built from other sources refuses the file
};
struct PlanFileRegion {
    int64_t bytes;
    int32_t kind, tag;
};
struct PlanFileReloc {
    int32_t field;            // index of the pointer field in for_each_pointer order
    int32_t region;
    int64_t offset;
};
}  // namespace ymi

extern "C" int ymi_plan_export(const ymi_plan* p, const ymi_plan_region* regions, int n_regions, const char* path, void* stream) {
    YMI_REQUIRE(p && regions && n_regions > 0 && path, "ymi_plan_export: null argument");
    for (int r = 0; r < n_regions; ++r)
        YMI_REQUIRE(regions[r].base && regions[r].bytes > 0 && regions[r].kind >= YMI_REGION_CONST && regions[r].kind <= YMI_REGION_IO, "ymi_plan_export: region %d is empty or of unknown kind", r);
    YMI_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));   // the constant regions are read back: whatever fills them has to be done
    FILE* f = fopen(path, "wb");
    YMI_REQUIRE(f != nullptr, "ymi_plan_export: cannot open %s for writing", path);
    PlanFileHeader h;
    memcpy(h.magic, "YMIPLAN1", 8);
    h.abi = YMI_ABI_VERSION; h.n_regions = n_regions; h.n_ops = (int)p->ops.size(); h.fuse_stem = p->fuse_stem ? 1 : 0; h.op_bytes = (int64_t)sizeof(Op);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    for (int r = 0; r < n_regions && ok; ++r) {
        PlanFileRegion fr = {regions[r].bytes, regions[r].kind, regions[r].tag};
        ok = fwrite(&fr, sizeof(fr), 1, f) == 1;
    }
    std::vector<unsigned char> host;
    for (int r = 0; r < n_regions && ok; ++r) {
        if (regions[r].kind != YMI_REGION_CONST) continue;
        host.resize((size_t)regions[r].bytes);
        if (hipMemcpy(host.data(), regions[r].base, host.size(), hipMemcpyDeviceToHost) != hipSuccess) {
            fclose(f);
            set_error("ymi_plan_export: reading region %d back failed", r);
            return YMI_EHIP;
        }
        ok = fwrite(host.data(), 1, host.size(), f) == host.size();
    }
    int rc = YMI_OK;
    for (size_t i = 0; i < p->ops.size() && ok && rc == YMI_OK; ++i) {
        Op op = p->ops[i];
        std::vector<PlanFileReloc> rel;
        int field = 0;
        for_each_pointer(op, [&](void** pp) {
            const char* v = (const char*)*pp;
            if (v != nullptr) {
                int hit = -1;
                for (int r = 0; r < n_regions && hit < 0; ++r) {
                    const char* b = (const char*)regions[r].base;
                    if (v >= b && v < b + regions[r].bytes) hit = r;
                }
                for (int r = 0; r < n_regions && hit < 0; ++r)   // (a one-past-the-end pointer, only when no region holds the address: allocations may be adjacent)
                    if (v == (const char*)regions[r].base + regions[r].bytes) hit = r;
                if (hit < 0) {
                    if (rc == YMI_OK) set_error("ymi_plan_export: op %d, pointer field %d (%p) lies in none of the %d regions", (int)i, field, (const void*)v, n_regions);
                    rc = YMI_EINVAL;
                } else {
                    rel.push_back({field, hit, (int64_t)(v - (const char*)regions[hit].base)});
                }
                *pp = nullptr;
            }
            ++field;
        });
        const int32_t nrel = (int32_t)rel.size();
        ok = fwrite(&op, sizeof(op), 1, f) == 1 && fwrite(&nrel, sizeof(nrel), 1, f) == 1 && (nrel == 0 || fwrite(rel.data(), sizeof(PlanFileReloc), rel.size(), f) == rel.size());
    }
    ok = (fclose(f) == 0) && ok;
    if (rc != YMI_OK) return rc;
    YMI_REQUIRE(ok, "ymi_plan_export: writing %s failed", path);
    return YMI_OK;
}

extern "C" int ymi_plan_import(const char* path, ymi_plan** out_plan, ymi_plan_region* regions_out, int max_regions, int* n_regions_out) {
    YMI_REQUIRE(path && out_plan, "ymi_plan_import: null argument");
    *out_plan = nullptr;
    FILE* f = fopen(path, "rb");
    YMI_REQUIRE(f != nullptr, "ymi_plan_import: cannot open %s", path);
    PlanFileHeader h;
    std::vector<PlanFileRegion> fr;
    ymi_plan* p = nullptr;
    std::vector<void*> base;
    auto fail = [&](const char* why) {
        fclose(f);
        if (p) ymi_plan_destroy(p);
        else for (void* m : base) (void)hipFree(m);
        set_error("ymi_plan_import: %s (%s)", why, path);
        return YMI_EINVAL;
    };
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "YMIPLAN1", 8) != 0) return fail("not a plan file");
    if (h.abi != YMI_ABI_VERSION || h.op_bytes != (int64_t)sizeof(Op)) return fail("written by another ABI version of the library");
    if (h.n_regions <= 0 || h.n_regions > (1 << 20) || h.n_ops < 0 || h.n_ops > (1 << 20)) return fail("corrupt header");
    fr.resize(h.n_regions);
    if (fread(fr.data(), sizeof(PlanFileRegion), fr.size(), f) != fr.size()) return fail("truncated region table");
    std::vector<unsigned char> host;
    for (int r = 0; r < h.n_regions; ++r) {
        if (fr[r].bytes <= 0) return fail("corrupt region table");
        // (slack behind every region: the exporting process carved its tensors out of an allocator's large segments, where a kernel's 16-byte chunk or padded weight
        // row that reaches a little past a tensor's end reads neighbouring memory; here every region is an allocation of its own)
        void* m = nullptr;
        constexpr size_t SLACK = 64 * 1024;
        if (hipMalloc(&m, (size_t)fr[r].bytes + SLACK) != hipSuccess) return fail("out of device memory");
        base.push_back(m);
        if (hipMemset((char*)m + fr[r].bytes, 0, SLACK) != hipSuccess) return fail("memset failed");
        if (fr[r].kind == YMI_REGION_CONST) {
            host.resize((size_t)fr[r].bytes);
            if (fread(host.data(), 1, host.size(), f) != host.size()) return fail("truncated region contents");
            if (hipMemcpy(m, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return fail("upload failed");
        } else if (hipMemset(m, 0, (size_t)fr[r].bytes) != hipSuccess) {   // scratch / IO regions start zeroed (zero tails, zero-initialised concat buffers)
            return fail("memset failed");
        }
    }
    p = ymi_plan_create();
    if (!p) return fail("out of memory");
    p->owned = base;
    p->fuse_stem = h.fuse_stem != 0;
    for (int i = 0; i < h.n_ops; ++i) {
        Op op;
        int32_t nrel = 0;
        if (fread(&op, sizeof(op), 1, f) != 1 || fread(&nrel, sizeof(nrel), 1, f) != 1 || nrel < 0 || nrel > 256) return fail("truncated op list");
        std::vector<PlanFileReloc> rel((size_t)nrel);
        if (nrel && fread(rel.data(), sizeof(PlanFileReloc), rel.size(), f) != rel.size()) return fail("truncated relocation list");
        int field = 0;
        size_t k = 0;
        bool bad = false;
        for_each_pointer(op, [&](void** pp) {
            *pp = nullptr;
            if (k < rel.size() && rel[k].field == field) {
                const PlanFileReloc& q = rel[k++];
                if (q.region < 0 || q.region >= h.n_regions || q.offset < 0 || q.offset > fr[q.region].bytes) bad = true;
                else *pp = (char*)base[q.region] + q.offset;
            }
            ++field;
        });
        if (bad || k != rel.size()) return fail("corrupt relocation");
        p->ops.push_back(op);
    }
    fclose(f);
    if (n_regions_out) *n_regions_out = h.n_regions;
    if (regions_out)
        for (int r = 0; r < h.n_regions && r < max_regions; ++r) {
            regions_out[r].base = base[r]; regions_out[r].bytes = fr[r].bytes; regions_out[r].kind = fr[r].kind; regions_out[r].tag = fr[r].tag;
        }
    *out_plan = p;
    return YMI_OK;
}
