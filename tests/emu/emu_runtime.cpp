// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/include/hip/hip_runtime.h.
#include <ucontext.h>

#include <vector>

#include <hip/hip_runtime.h>

namespace emu {

ThreadCtx* cur = nullptr;
const void* kernarg_ptr = nullptr;
std::vector<std::function<void()>>* capture = nullptr;

namespace {

enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    ThreadCtx tc;
    State state = DONE;
    char* stack = nullptr;
};

constexpr size_t kStack = 96 * 1024;
constexpr int kWaveSize = 64;

std::vector<Fiber> fibers;
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

uint64_t peer_slot(int mask, bool* valid) {
    const int lane = cur_idx % kWaveSize;
    const int src = (cur_idx - lane) + (lane ^ mask);
    if (src < 0 || src >= (int)fibers.size() || (lane ^ mask) >= kWaveSize) { *valid = false; return 0; }
    *valid = true;
    return fibers[src].tc.slot;
}

uint64_t peer_slot_lane(int src_lane, bool* valid) {
    const int lane = cur_idx % kWaveSize;
    const int src = (cur_idx - lane) + src_lane;
    if (src_lane < 0 || src_lane >= kWaveSize || src >= (int)fibers.size()) { *valid = false; return 0; }
    *valid = true;
    return fibers[src].tc.slot;
}

uint64_t peer_rl(int src_lane, bool* valid) {
    const int lane = cur_idx % kWaveSize;
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
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = fibers[t];
            f.tc.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.tc.bid = dim3(bx, by, bz);
            f.tc.bdim = block;
            f.tc.gdim = grid;
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

}  // namespace emu
