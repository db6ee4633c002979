// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/include/hip/hip_runtime.h.
#include <ucontext.h>

#include <vector>

#include <hip/hip_runtime.h>

namespace emu {

ThreadCtx* cur = nullptr;
const void* kernarg_ptr = nullptr;
std::vector<std::function<void()>>* capture = nullptr;

namespace {

enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3, YIELDED = 4 };

struct Fiber {
    ucontext_t ctx;
    ThreadCtx tc;
    State state = DONE;
    char* stack = nullptr;
};

constexpr size_t kStack = 96 * 1024;
constexpr int kWaveSize = 64;

std::vector<Fiber> fibers;
int base_idx = 0;                 // first fiber of the running block (concurrent launches)
bool concurrent = false;
ucontext_t sched_ctx;
const std::function<void()>* body_ptr = nullptr;
int cur_idx = -1;

void trampoline() {
    (*body_ptr)();
    fibers[cur_idx].state = DONE;
    swapcontext(&fibers[cur_idx].ctx, &sched_ctx);
}

void yield_with(State s) {
    fibers[cur_idx].state = s;
    swapcontext(&fibers[cur_idx].ctx, &sched_ctx);
}

}  // namespace

void sync_block() { yield_with(WAIT_BLOCK); }
void sync_wave() { yield_with(WAIT_WAVE); }
void poll_yield() { if (concurrent) yield_with(YIELDED); }
// Only in launches that keep their LDS state in the per-workgroup dynamic buffer: `__shared__` variables are plain statics
// here (one copy for all workgroups), which is fine as long as a workgroup is only switched out where none of them is live
// (the s_sleep of a dependency wait) -- le_sweep_kernel; le_resident_kernel uses the dynamic buffer and can be switched
// out anywhere.
bool preempt_at_loads = false;
void visibility_point() {
    static uint32_t lcg = 12345u;
    if (!concurrent || !preempt_at_loads) return;
    lcg = lcg * 1664525u + 1013904223u;
    if ((lcg >> 16) % 4u == 0u) yield_with(YIELDED);
}

uint64_t peer_slot(int mask, bool* valid) {
    const int lane = (cur_idx - base_idx) % kWaveSize;
    const int src = (cur_idx - lane) + (lane ^ mask);
    if (src < 0 || src >= (int)fibers.size() || (lane ^ mask) >= kWaveSize) { *valid = false; return 0; }
    *valid = true;
    return fibers[src].tc.slot;
}

uint64_t peer_slot_lane(int src_lane, bool* valid) {
    const int lane = (cur_idx - base_idx) % kWaveSize;
    const int src = (cur_idx - lane) + src_lane;
    if (src_lane < 0 || src_lane >= kWaveSize || src >= (int)fibers.size()) { *valid = false; return 0; }
    *valid = true;
    return fibers[src].tc.slot;
}

uint64_t peer_rl(int src_lane, bool* valid) {
    const int lane = (cur_idx - base_idx) % kWaveSize;
    const int src = (cur_idx - lane) + src_lane;
    if (src_lane < 0 || src_lane >= kWaveSize || src >= (int)fibers.size() || !fibers[src].tc.rl_valid) { *valid = false; return 0; }
    *valid = true;
    return fibers[src].tc.rl_val;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || grid.x * grid.y * grid.z == 0) return;
    if ((int)fibers.size() < nthreads) {
        const size_t old = fibers.size();
        fibers.resize(nthreads);
        for (size_t i = old; i < fibers.size(); ++i) fibers[i].stack = (char*)malloc(kStack);
    }
    body_ptr = &body;
    // dynamic shared memory of a sequential launch: one buffer serves every workgroup in turn (an LDS's worth)
    static unsigned char* seq_smem = (unsigned char*)aligned_alloc(256, 160 * 1024);
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = fibers[t];
            f.tc.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.tc.bid = dim3(bx, by, bz);
            f.tc.bdim = block;
            f.tc.gdim = grid;
            f.tc.smem = seq_smem;
            f.tc.slot = 0;
            f.tc.rl_valid = false;
            f.state = READY;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = kStack;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, trampoline, 0);
        }
        for (;;) {
            bool ran = false;
            int live = 0;
            for (int t = 0; t < nthreads; ++t) {
                if (fibers[t].state == READY) {
                    cur_idx = t;
                    cur = &fibers[t].tc;
                    swapcontext(&sched_ctx, &fibers[t].ctx);
                    ran = true;
                }
                if (fibers[t].state != DONE) ++live;
            }
            if (live == 0) break;
            // release wave rendezvous: every live lane of the wave waits on the wave barrier
            bool released = false;
            for (int w = 0; w * kWaveSize < nthreads; ++w) {
                int waiting = 0, alive = 0;
                const int b = w * kWaveSize, e = std::min(nthreads, b + kWaveSize);
                for (int t = b; t < e; ++t) {
                    if (fibers[t].state != DONE) ++alive;
                    if (fibers[t].state == WAIT_WAVE) ++waiting;
                }
                if (alive > 0 && waiting == alive) {
                    for (int t = b; t < e; ++t) if (fibers[t].state == WAIT_WAVE) fibers[t].state = READY;
                    released = true;
                }
            }
            // release the block barrier: every live thread waits on it
            int wb = 0;
            for (int t = 0; t < nthreads; ++t) if (fibers[t].state == WAIT_BLOCK) ++wb;
            if (wb == live && live > 0) {
                for (int t = 0; t < nthreads; ++t) if (fibers[t].state == WAIT_BLOCK) fibers[t].state = READY;
                released = true;
            }
            if (!ran && !released) {
                fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier or shuffle\n", bx, by, bz);
                abort();
            }
        }
    }
    cur = nullptr;
    cur_idx = -1;
}

// ---- concurrent launch: all workgroups alive, round-robin over blocks ------------------------------------------
namespace {
constexpr size_t kSmallStack = 48 * 1024;
std::vector<Fiber> cfibers;
std::vector<std::vector<unsigned char>> csmem;

// run block b until every live fiber of it is blocked (barrier with absentees impossible here: a poll-yield of one
// thread while the others sit at the barrier is the normal shape of a wait) -- returns true if anything happened
bool run_block(int b, int nthreads) {
    bool progress = false;
    const int b0 = b * nthreads;
    base_idx = b0;
    for (;;) {
        bool ran = false;
        int live = 0;
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = fibers[b0 + t];
            if (f.state == READY) {
                cur_idx = b0 + t;
                cur = &f.tc;
                swapcontext(&sched_ctx, &f.ctx);
                ran = true;
            }
            if (f.state != DONE) ++live;
        }
        if (ran) progress = true;
        if (live == 0) return progress;
        bool released = false;
        for (int w = 0; w * kWaveSize < nthreads; ++w) {
            int waiting = 0, alive = 0;
            const int wb = b0 + w * kWaveSize, we = b0 + std::min(nthreads, (w + 1) * kWaveSize);
            for (int t = wb; t < we; ++t) {
                if (fibers[t].state != DONE) ++alive;
                if (fibers[t].state == WAIT_WAVE) ++waiting;
            }
            if (alive > 0 && waiting == alive) {
                for (int t = wb; t < we; ++t) if (fibers[t].state == WAIT_WAVE) fibers[t].state = READY;
                released = true;
            }
        }
        int wbk = 0;
        for (int t = 0; t < nthreads; ++t) if (fibers[b0 + t].state == WAIT_BLOCK) ++wbk;
        if (wbk == live) {
            for (int t = 0; t < nthreads; ++t) if (fibers[b0 + t].state == WAIT_BLOCK) fibers[b0 + t].state = READY;
            released = true;
        }
        if (released) { progress = true; continue; }
        if (!ran) return progress;       // only yielded pollers (and threads waiting for them) are left: next block
    }
}
}  // namespace

void launch_concurrent(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    if (nthreads <= 0 || nblocks <= 0) return;
    std::vector<Fiber> saved;
    saved.swap(fibers);                      // the sequential pool is left alone
    if (cfibers.size() < (size_t)nthreads * nblocks) {
        const size_t old = cfibers.size();
        cfibers.resize((size_t)nthreads * nblocks);
        for (size_t i = old; i < cfibers.size(); ++i) cfibers[i].stack = (char*)malloc(kSmallStack);
    }
    fibers.swap(cfibers);
    csmem.assign(nblocks, std::vector<unsigned char>(smem_bytes + 64, 0xA5));
    body_ptr = &body;
    concurrent = true;
    preempt_at_loads = smem_bytes > 0;
    for (int b = 0; b < nblocks; ++b) {
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = fibers[(size_t)b * nthreads + t];
            f.tc.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.tc.bid = dim3(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
            f.tc.bdim = block;
            f.tc.gdim = grid;
            f.tc.smem = csmem[b].data();
            f.tc.slot = 0;
            f.tc.rl_valid = false;
            f.state = READY;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = kSmallStack;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, trampoline, 0);
        }
    }
    long idle_rounds = 0;
    for (;;) {
        bool progress = false;
        int live_blocks = 0;
        for (int b = 0; b < nblocks; ++b) {
            // yielded pollers get another turn every round
            bool any_live = false;
            for (int t = 0; t < nthreads; ++t) {
                Fiber& f = fibers[(size_t)b * nthreads + t];
                if (f.state == YIELDED) f.state = READY;
                if (f.state != DONE) any_live = true;
            }
            if (!any_live) continue;
            ++live_blocks;
            // a round that only re-polls is not progress: run_block reports a poll as progress only via state changes
            int before = 0;
            for (int t = 0; t < nthreads; ++t) before += (int)fibers[(size_t)b * nthreads + t].state * (t + 1);
            run_block(b, nthreads);
            int after = 0;
            for (int t = 0; t < nthreads; ++t) after += (int)fibers[(size_t)b * nthreads + t].state * (t + 1);
            if (before != after) progress = true;
        }
        if (live_blocks == 0) break;
        idle_rounds = progress ? 0 : idle_rounds + 1;
        if (idle_rounds > 50000000L) {       // the kernels' own spin limits fire long before this
            fprintf(stderr, "emu: concurrent launch made no progress (deadlock between workgroups)\n");
            abort();
        }
    }
    concurrent = false;
    preempt_at_loads = false;
    base_idx = 0;
    fibers.swap(cfibers);
    fibers.swap(saved);
    cur = nullptr;
    cur_idx = -1;
}

}  // namespace emu
