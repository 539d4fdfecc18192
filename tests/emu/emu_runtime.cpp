// TEST INFRASTRUCTURE ONLY -- see emu_runtime.h.
#include "emu_runtime.h"

#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

namespace emu {
namespace {
constexpr size_t kStack = 512 * 1024;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
};

struct Wave {
    float a[64], b[64];
    unsigned short a16[64][8], b16[64][8];
    unsigned short t16[64][4];
    int arrived = 0;
    unsigned gen = 0;
};

ucontext_t g_sched;
std::vector<Fiber> g_fibers;
std::vector<Wave> g_waves;
const std::function<void()>* g_body = nullptr;
int g_cur = 0, g_block = 0, g_bid = 0, g_grid = 0;
int g_barrier_count = 0;
unsigned g_barrier_gen = 0;
float* g_smem = nullptr;
unsigned long long g_shuffle = 0, g_rng = 0;

unsigned next_random() {                 // xorshift64*
    g_rng ^= g_rng >> 12; g_rng ^= g_rng << 25; g_rng ^= g_rng >> 27;
    return (unsigned)((g_rng * 2685821657736338717ull) >> 33);
}

void yield() { swapcontext(&g_fibers[g_cur].ctx, &g_sched); }

void trampoline() {
    (*g_body)();
    g_fibers[g_cur].done = true;
    swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}

// rendezvous of the 64 lanes of the calling wave; returns after every lane has arrived.
void wave_sync(Wave& w) {
    unsigned my = w.gen;
    if (++w.arrived == 64) { w.arrived = 0; ++w.gen; return; }
    while (w.gen == my) yield();
}
}  // namespace

int tid() { return g_cur; }
int bid() { return g_bid; }
int nblk() { return g_grid; }
float* smem() { return g_smem; }

void sync_block() {
    unsigned my = g_barrier_gen;
    if (++g_barrier_count == g_block) { g_barrier_count = 0; ++g_barrier_gen; return; }
    while (g_barrier_gen == my) yield();
}

void sync_wave() { wave_sync(g_waves[g_cur >> 6]); }
void yield_fiber() { yield(); }

f32x4 mfma16(float a, float b, f32x4 c) {
    Wave& w = g_waves[g_cur >> 6];
    const int l = g_cur & 63;
    w.a[l] = a; w.b[l] = b;
    wave_sync(w);
    const int col = l & 15, rq = l >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = rq * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.a[k * 16 + row], w.b[k * 16 + col], acc);
        d[r] = acc;
    }
    wave_sync(w);      // nobody overwrites a/b before everyone has read them
    return d;
}

static inline double bf16_value(unsigned short h) {
    const unsigned bits = (unsigned)h << 16;
    float f;
    memcpy(&f, &bits, 4);
    return (double)f;
}

f32x4 mfma16_bf16(const unsigned short* a8, const unsigned short* b8, f32x4 c) {
    Wave& w = g_waves[g_cur >> 6];
    const int l = g_cur & 63;
    for (int e = 0; e < 8; ++e) { w.a16[l][e] = a8[e]; w.b16[l][e] = b8[e]; }
    wave_sync(w);
    const int col = l & 15, rq = l >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = rq * 4 + r;
        double acc = (double)c[r];
        for (int g = 0; g < 4; ++g)                 // k slot 8 g + e: lane (row, g) of A, lane (col, g) of B
            for (int e = 0; e < 8; ++e) acc += bf16_value(w.a16[g * 16 + row][e]) * bf16_value(w.b16[g * 16 + col][e]);
        d[r] = (float)acc;
    }
    wave_sync(w);
    return d;
}

void lds_tr16(const void* p, unsigned short* out4) {
    Wave& w = g_waves[g_cur >> 6];
    const int l = g_cur & 63, grp = l & ~15, n = l & 15;
    if (((uintptr_t)p & 7) != 0) { fprintf(stderr, "emu: ds_read_b64_tr_b16 address not 8-byte aligned\n"); abort(); }
    memcpy(w.t16[l], p, 8);
    wave_sync(w);
    for (int j = 0; j < 4; ++j) out4[j] = w.t16[grp + 4 * j + n / 4][n % 4];     // element (row j, column n) of the group's block
    wave_sync(w);
}

float shfl_xor(float v, int mask) {
    Wave& w = g_waves[g_cur >> 6];
    const int l = g_cur & 63;
    w.a[l] = v;
    wave_sync(w);
    float r = w.a[(l ^ mask) & 63];
    wave_sync(w);
    return r;
}

float row_sum16(float v) {
    // same butterfly as the DPP sequence in pinn_port.h: xor 1, xor 2, half-mirror (i <-> 7-i), mirror (i <-> 15-i)
    Wave& w = g_waves[g_cur >> 6];
    const int l = g_cur & 63, row = l & ~15, i = l & 15;
    const int partner[4] = {i ^ 1, i ^ 2, (i & 8) | (7 - (i & 7)), 15 - i};
    for (int step = 0; step < 4; ++step) {
        w.a[l] = v;
        wave_sync(w);
        const float o = w.a[row + partner[step]];
        wave_sync(w);
        v += o;
    }
    return v;
}

void launch(int grid, int block, size_t smem_bytes, const std::function<void()>& body) {
    if (block % 64) { fprintf(stderr, "emu: block size must be a multiple of 64\n"); abort(); }
    const char* sh = getenv("PINN_EMU_SHUFFLE");
    g_shuffle = sh ? strtoull(sh, nullptr, 10) : 0;
    if (g_shuffle && !g_rng) g_rng = g_shuffle * 0x9E3779B97F4A7C15ull + 1;
    g_body = &body; g_block = block; g_grid = grid;
    g_fibers.assign(block, Fiber());
    for (auto& f : g_fibers) f.stack = (char*)malloc(kStack);
    std::vector<float> lds(smem_bytes / 4 + 4);
    g_smem = lds.data();
    for (int b = 0; b < grid; ++b) {
        g_bid = b; g_barrier_count = 0;
        g_waves.assign(block / 64, Wave());
        for (int i = 0; i < block; ++i) {
            Fiber& f = g_fibers[i];
            f.done = false;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = &g_sched;
            makecontext(&f.ctx, trampoline, 0);
        }
        int remaining = block;
        const int n_waves = block / 64;
        std::vector<int> order(n_waves), stall(n_waves, 0);
        for (int w = 0; w < n_waves; ++w) order[w] = w;
        while (remaining > 0) {
            remaining = 0;
            if (g_shuffle) {
                // PINN_EMU_SHUFFLE=<seed>: the waves of a workgroup advance in a random order, and now and then a wave STALLS for a random
                // stretch of up to 512 scheduling passes (a pass moves a wave from one wave-wide rendezvous -- an MFMA, a shuffle -- to
                // the next): the others run whole phases ahead of it, up to the next barrier, so that a missing barrier (a fast wave
                // overwriting LDS a slow one still reads) changes results instead of hiding behind round-robin
                for (int w = n_waves - 1; w > 0; --w) { const int r = (int)(next_random() % (unsigned)(w + 1)); std::swap(order[w], order[r]); }
                bool all_stalled = true;
                for (int w = 0; w < n_waves; ++w) {
                    if (stall[w] > 0) --stall[w];
                    else if ((next_random() & 127) == 0) stall[w] = 1 + (int)(next_random() & 511);
                    if (stall[w] == 0) all_stalled = false;
                }
                if (all_stalled) std::fill(stall.begin(), stall.end(), 0);
            }
            for (int wi = 0; wi < n_waves; ++wi) {
                const int w = order[wi];
                if (g_shuffle && stall[w] > 0) continue;
                for (int i = 64 * w; i < 64 * w + 64; ++i) {
                    if (g_fibers[i].done) continue;
                    g_cur = i;
                    swapcontext(&g_sched, &g_fibers[i].ctx);
                }
            }
            for (int i = 0; i < block; ++i)
                if (!g_fibers[i].done) ++remaining;
        }
    }
    for (auto& f : g_fibers) free(f.stack);
    g_fibers.clear();
}
}  // namespace emu
