// hipsim runtime: fibers, barriers, the per-wave exchange space (see hipsim.h).  Compiled WITHOUT the keyword macros of hipsim.h
// in effect for its own body (it is included first, before the kernel sources, and uses none of them).
#include <ucontext.h>

#include <memory>

namespace hipsim {

dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
int g_max_lds = 0;
long g_launches = 0;
const void* g_kernarg = nullptr;

namespace {
constexpr size_t STACK = 256 * 1024;
struct Bar {
    int count = 0, n = 0;
    unsigned gen = 0;
};
struct Lane {
    ucontext_t ctx;
    std::unique_ptr<char[]> stack;
    bool done = false;
};
struct Block {
    std::vector<Lane> lanes;
    std::vector<Bar> wave_bar;
    std::vector<std::vector<unsigned char>> dep;   // per wave: 64 x 64 bytes
    Bar block_bar;
    ucontext_t sched;
    int cur = -1;
    bool progress = false;
    const std::function<void()>* body = nullptr;
};
Block* g_blk = nullptr;

void trampoline() {
    Block* b = g_blk;
    (*b->body)();
    b->lanes[b->cur].done = true;
    b->progress = true;
    swapcontext(&b->lanes[b->cur].ctx, &b->sched);
}

void bar_wait(Bar& bar) {
    const unsigned g = bar.gen;
    if (++bar.count == bar.n) {
        bar.count = 0;
        ++bar.gen;
        g_blk->progress = true;
    } else {
        while (bar.gen == g) yield();
    }
}
}  // namespace

void yield() {
    Block* b = g_blk;
    swapcontext(&b->lanes[b->cur].ctx, &b->sched);
}
void wave_sync() { bar_wait(g_blk->wave_bar[g_blk->cur >> 6]); }
void block_sync() { bar_wait(g_blk->block_bar); }
unsigned char* wave_deposit() { return g_blk->dep[g_blk->cur >> 6].data(); }

void run_block(const std::function<void()>& body, dim3 grid, dim3 block, dim3 bidx) {
    const int n = (int)block.x;
    if (n % 64 != 0 || block.y != 1 || block.z != 1) {
        fprintf(stderr, "hipsim: 1-D blocks of whole waves only\n");
        abort();
    }
    Block b;
    b.lanes.resize(n);
    b.wave_bar.resize(n / 64);
    for (auto& w : b.wave_bar) w.n = 64;
    b.dep.assign(n / 64, std::vector<unsigned char>(64 * 64));
    b.block_bar.n = n;
    b.body = &body;
    g_blk = &b;
    g_blockIdx = bidx;
    g_blockDim = block;
    g_gridDim = grid;
    for (int i = 0; i < n; ++i) {
        Lane& l = b.lanes[i];
        l.stack.reset(new char[STACK]);
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack.get();
        l.ctx.uc_stack.ss_size = STACK;
        l.ctx.uc_link = nullptr;
        makecontext(&l.ctx, trampoline, 0);
    }
    int ndone = 0;
    while (ndone < n) {
        b.progress = false;
        ndone = 0;
        static const bool reverse = getenv("HIPSIM_REVERSE") != nullptr;   // debugging aid: schedule the lanes of a block in descending order
        for (int ii = 0; ii < n; ++ii) {
            const int i = reverse ? n - 1 - ii : ii;
            if (b.lanes[i].done) {
                ++ndone;
                continue;
            }
            b.cur = i;
            g_threadIdx = dim3((unsigned)i);
            swapcontext(&b.sched, &b.lanes[i].ctx);
            if (b.lanes[i].done) ++ndone;
        }
        if (!b.progress && ndone < n) {
            fprintf(stderr, "hipsim: DEADLOCK in block (%u, %u, %u) -- lanes wait at barriers not every lane reaches (divergent __syncthreads or wave collective)\n", bidx.x, bidx.y, bidx.z);
            abort();
        }
    }
    g_blk = nullptr;
}

}  // namespace hipsim
