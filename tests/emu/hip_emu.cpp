// hip_emu.cpp -- fiber scheduler behind hip_emu.hpp (CPU test infrastructure only).
#include "hip_emu.hpp"

#include <algorithm>
#include <limits>
#include <map>
#include <vector>

namespace emu {

dim3 g_block, g_bdim, g_gdim;
Lane *g_cur = nullptr;

namespace {

enum State { RUNNABLE, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Access {
    int site, occ;
    uintptr_t addr;
    bool write;
};

struct Fiber {
    Lane lane;
    void *sp = nullptr;
    char *stack = nullptr;
    State state = RUNNABLE;
    int index = 0;
    int xparity = 0;                 // alternates the exchange buffer (one rendez-vous per collective)
    std::vector<Access> acc;
    std::map<int, int> occ;
};

constexpr size_t kStack = 256 * 1024;
std::vector<Fiber> fibers;
std::vector<uint64_t> xbuf;          // [wave][parity][64]
std::vector<unsigned char> xwide;    // [wave][parity][64][32]
void *sched_sp = nullptr;
Fiber *cur_fiber = nullptr;
const std::function<void()> *cur_body = nullptr;
Stats g_stats = {};
void *lds_base = nullptr;
size_t lds_bytes = 0;
bool lds_trace = false;

extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

void yield_to_scheduler() { emu_switch(&cur_fiber->sp, sched_sp); }

void fiber_entry() {
    (*cur_body)();
    cur_fiber->state = DONE;
    yield_to_scheduler();
    std::fprintf(stderr, "emu: resumed a finished fiber\n");
    std::abort();
}

void prepare(Fiber &f) {
    if (!f.stack) f.stack = static_cast<char *>(std::malloc(kStack));
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
    void **s = reinterpret_cast<void **>(top);
    s[-1] = nullptr;                                   // fake return address of fiber_entry
    s[-2] = reinterpret_cast<void *>(&fiber_entry);    // popped by emu_switch's ret
    for (int i = 3; i <= 8; ++i) s[-i] = nullptr;      // rbp rbx r12 r13 r14 r15
    f.sp = &s[-8];
    f.state = RUNNABLE;
    f.xparity = 0;
    f.acc.clear();
    f.occ.clear();
}

void analyse_lds(int nthreads) {
    if (!lds_trace) return;
    int nwaves = (nthreads + 63) / 64;
    for (int w = 0; w < nwaves; ++w) {
        std::map<std::pair<int, int>, std::vector<std::pair<int, uintptr_t>>> groups[2];
        for (int l = 0; l < 64 && w * 64 + l < nthreads; ++l)
            for (const Access &a : fibers[w * 64 + l].acc)
                groups[a.write][{a.site, a.occ}].push_back({l, a.addr});
        for (int wr = 0; wr < 2; ++wr)
            for (auto &kv : groups[wr]) {
                unsigned cycles = 0;
                for (int half = 0; half < 2; ++half) {
                    std::map<unsigned, std::vector<uintptr_t>> banks;
                    for (auto &la : kv.second)
                        if (la.first / 32 == half) {
                            auto &v = banks[(la.second / 4) % 32];
                            if (std::find(v.begin(), v.end(), la.second) == v.end()) v.push_back(la.second);
                        }
                    unsigned worst = 0;
                    for (auto &b : banks) worst = std::max<unsigned>(worst, b.second.size());
                    cycles += worst;
                }
                if (wr) { g_stats.lds_write_instr++; g_stats.lds_write_cycles += cycles; }
                else    { g_stats.lds_read_instr++;  g_stats.lds_read_cycles += cycles; }
            }
    }
}

void run_block(int nthreads) {
    int nwaves = (nthreads + 63) / 64;
    xbuf.assign(size_t(nwaves) * 2 * 64, 0);
    xwide.assign(size_t(nwaves) * 2 * 64 * 32, 0);
    for (int t = 0; t < nthreads; ++t) prepare(fibers[t]);
    for (;;) {
        bool progressed = false, all_done = true;
        for (int t = 0; t < nthreads; ++t) {
            Fiber &f = fibers[t];
            if (f.state == DONE) continue;
            all_done = false;
            if (f.state != RUNNABLE) continue;
            cur_fiber = &f;
            g_cur = &f.lane;
            emu_switch(&sched_sp, f.sp);
            progressed = true;
        }
        if (all_done) break;
        // release wave rendez-vous
        for (int w = 0; w < nwaves; ++w) {
            int lo = w * 64, hi = std::min(nthreads, lo + 64), waiting = 0, alive = 0;
            for (int t = lo; t < hi; ++t) {
                if (fibers[t].state != DONE) ++alive;
                if (fibers[t].state == WAIT_WAVE) ++waiting;
            }
            if (alive && waiting == alive) {
                for (int t = lo; t < hi; ++t)
                    if (fibers[t].state == WAIT_WAVE) fibers[t].state = RUNNABLE;
                progressed = true;
            }
        }
        // release block barrier
        int waiting = 0, alive = 0;
        for (int t = 0; t < nthreads; ++t) {
            if (fibers[t].state != DONE) ++alive;
            if (fibers[t].state == WAIT_BLOCK) ++waiting;
        }
        if (alive && waiting == alive) {
            for (int t = 0; t < nthreads; ++t) fibers[t].state = RUNNABLE;
            progressed = true;
        }
        if (!progressed) {
            std::fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier or collective\n",
                         g_block.x, g_block.y, g_block.z);
            std::abort();
        }
    }
    analyse_lds(nthreads);
}

}  // namespace

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    const char *env = std::getenv("CCA_EMU_LDS");
    lds_trace = env && env[0] == '1';
    g_gdim = grid;
    g_bdim = block;
    int nthreads = int(block.x * block.y * block.z);
    if (int(fibers.size()) < nthreads) fibers.resize(nthreads);
    for (int t = 0; t < nthreads; ++t) {
        fibers[t].index = t;
        fibers[t].lane.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    }
    cur_body = &body;
    g_stats.launches++;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                g_block = dim3(x, y, z);
                lds_base = nullptr;
                lds_bytes = 0;
                run_block(nthreads);
            }
    cur_body = nullptr;
}

void block_barrier() {
    cur_fiber->state = WAIT_BLOCK;
    yield_to_scheduler();
}

int lane_id() { return cur_fiber->index & 63; }

const uint64_t *wave_exchange(uint64_t mine) {
    Fiber &f = *cur_fiber;
    int wave = f.index / 64;
    uint64_t *slots = &xbuf[(size_t(wave) * 2 + f.xparity) * 64];
    f.xparity ^= 1;
    slots[f.index & 63] = mine;
    f.state = WAIT_WAVE;
    yield_to_scheduler();
    return slots;
}

const unsigned char *wave_exchange_bytes(const void *mine, size_t nbytes) {
    Fiber &f = *cur_fiber;
    int wave = f.index / 64;
    unsigned char *slots = &xwide[(size_t(wave) * 2 + f.xparity) * 64 * 32];
    f.xparity ^= 1;
    std::memcpy(slots + size_t(f.index & 63) * 32, mine, nbytes <= 32 ? nbytes : 32);
    f.state = WAIT_WAVE;
    yield_to_scheduler();
    return slots;
}

void lds_register(void *base, size_t bytes) {
    // first caller of the block wins; it runs before any other fiber touches LDS, and the
    // kernel follows the registration with a __syncthreads().
    // (a kernel may register several __shared__ arrays: each is poisoned once per block, by its first caller)
    static void *seen[8];
    static int nseen = 0;
    if (!lds_base) nseen = 0;                                // first registration of this block
    for (int i = 0; i < nseen; ++i)
        if (seen[i] == base) return;
    if (nseen < 8) seen[nseen++] = base;
    if (!lds_base) {
        lds_base = base;
        lds_bytes = bytes;
    }
    uint32_t *p = static_cast<uint32_t *>(base);
    for (size_t i = 0; i < bytes / 4; ++i) p[i] = 0x7fc0dead;
}

static void note(const void *addr, int site, bool write) {
    if (!lds_trace) return;
    Fiber &f = *cur_fiber;
    int occ = f.occ[site * 2 + write]++;
    f.acc.push_back({site, occ, reinterpret_cast<uintptr_t>(addr), write});
}
void lds_note_read(const void *addr, int site) { note(addr, site, false); }
void lds_note_write(const void *addr, int site) { note(addr, site, true); }

Stats &stats() { return g_stats; }

}  // namespace emu

// emulator-only exports (not part of include/ccnet_cca.h): counters for the tests
extern "C" void cca_emu_stats(unsigned long long *out6) {
    const emu::Stats &s = emu::stats();
    out6[0] = s.lds_read_instr; out6[1] = s.lds_read_cycles; out6[2] = s.lds_write_instr;
    out6[3] = s.lds_write_cycles; out6[4] = s.mfma; out6[5] = s.launches;
}
extern "C" void cca_emu_reset_stats() { emu::stats() = emu::Stats{}; }
