// tools/microbench.hip -- developer micro-benchmark for the step kernel (not part of the product).
// Times kernel VARIANTS built from the same board_ops.hpp with hipEvents so that a design choice
// (boards per lane, block size, persistent grid, RNG cost, memory-only floor) is measured, not
// guessed.  Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Isimulate_2048_amd/csrc
//                         tools/microbench.hip -o tools/microbench
// Run on the GPU box:  ./tools/microbench [log2_boards=20] [launches=300]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>
#include <algorithm>
#include <functional>
#include <random>
#include <chrono>
#include <dlfcn.h>

#include "env2048.h"
#include "board_ops.hpp"
#include "microbench_v2.hpp"

using namespace env2048;

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

struct Args {
    uint64_t n;
    uint4 *boards;
    const uint8_t *actions;
    int32_t *reward;
    uint8_t *done;
    RngKey key;
    uint32_t flags;
};

__device__ __forceinline__ Board ld(const uint4 *p, uint64_t i)
{
    uint4 v = p[i];
    return Board{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st(uint4 *p, uint64_t i, const Board &b) { p[i] = make_uint4(b.r0, b.r1, b.r2, b.r3); }
// round-1 "v1" transition: cndmask direction handling + two-word spawn (kept here as the baseline)
__device__ __forceinline__ void transition_v1(Board &bd, uint32_t action, uint32_t w_cell, uint32_t w_val, uint32_t &reward,
                                              bool &changed, bool &done)
{
    afterstate(bd, action & 3u, reward, changed);
    uint32_t empties_left = spawn_tile(bd, w_cell, w_val, changed);
    done = (empties_left == 0u) && no_equal_neighbours(bd);
}

// ---- V1: the product kernel's shape (1 board per lane, grid = n/256) -------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_v1(Args a)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    Board bd = ld(a.boards, i);
    uint32_t w0, w1, w2, w3;
    env_block(a.key, i, 0u, w0, w1, w2, w3);
    uint32_t reward;
    bool changed, done;
    transition_v1(bd, a.actions[i], w0, w1, reward, changed, done);
    if ((a.flags & 1u) && done) {
        env_block(a.key, i, 1u, w0, w1, w2, w3);
        bd = fresh_board(w0, w1, w2, w3);
    }
    st(a.boards, i, bd);
    a.reward[i] = (int32_t)reward;
    a.done[i] = done ? 1 : 0;
}

// ---- memory-only floor: same accesses, trivial arithmetic ------------------------------------
__global__ __launch_bounds__(256) void k_memonly(Args a)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    Board bd = ld(a.boards, i);
    uint32_t act = a.actions[i];
    bd.r0 ^= act;
    st(a.boards, i, bd);
    a.reward[i] = (int32_t)(bd.r1 & 0xFF);
    a.done[i] = (uint8_t)(bd.r2 & 1);
}

// ---- memory-only floor with write-through (sc1) wide stores, 4 boards per lane like the product ------
typedef unsigned int mb_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_memonly_sc1(Args a)
{
    const uint64_t lane_id = (uint64_t)blockIdx.x * 512 + threadIdx.x;
    const uint64_t tile0 = (lane_id & ~63ull) * 4;
    const uint32_t lane = threadIdx.x & 63u;
    if (tile0 >= a.n) return;
    uint4 v[4];
    uint32_t act = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = a.boards[tile0 + 64 * j + lane];
        act ^= a.actions[tile0 + 64 * j + lane];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mb_u32x4 w = {v[j].x ^ act, v[j].y, v[j].z, v[j].w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(a.boards + tile0 + 64 * j + lane), "v"(w) : "memory");
    }
    mb_u32x4 r = {v[0].y & 0xFF, v[1].y & 0xFF, v[2].y & 0xFF, v[3].y & 0xFF};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(a.reward + tile0 + 4 * lane), "v"(r) : "memory");
    uint32_t d = (v[0].z & 1) | ((v[1].z & 1) << 8) | ((v[2].z & 1) << 16) | ((v[3].z & 1) << 24);
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(a.done + tile0 + 4 * lane), "v"(d) : "memory");
}

// ---- no-Philox: cheap mixing instead of the 10-round generator (NOT the contract; cost probe) -
__global__ __launch_bounds__(256) void k_nophilox(Args a)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    Board bd = ld(a.boards, i);
    uint32_t w0 = ((uint32_t)i * 0x9E3779B9u) ^ a.key.step_lo, w1 = (w0 >> 7) ^ (w0 * 0x85EBCA6Bu);
    uint32_t reward;
    bool changed, done;
    transition_v1(bd, a.actions[i], w0, w1, reward, changed, done);
    if ((a.flags & 1u) && done) bd = fresh_board(w0, w1, ~w0, ~w1);
    st(a.boards, i, bd);
    a.reward[i] = (int32_t)reward;
    a.done[i] = done ? 1 : 0;
}

// ---- Philox only (no game logic): RNG cost probe ----------------------------------------------
__global__ __launch_bounds__(256) void k_philoxonly(Args a)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    Board bd = ld(a.boards, i);
    uint32_t w0, w1, w2, w3;
    env_block(a.key, i, 0u, w0, w1, w2, w3);
    bd.r0 ^= w0 & 1;
    bd.r1 ^= w1 & 1;
    st(a.boards, i, bd);
    a.reward[i] = (int32_t)(w2 & 3);
    a.done[i] = (uint8_t)(w3 & 1);
}

// ---- multi-board per lane: B boards per thread, wave-strided so every access stays coalesced --
template <int B>
__global__ __launch_bounds__(256) void k_multi(Args a)
{
    const uint64_t base = ((uint64_t)blockIdx.x * 256) * B + threadIdx.x;
    Board bd[B];
    uint32_t act[B];
#pragma unroll
    for (int j = 0; j < B; ++j) {
        uint64_t i = base + (uint64_t)j * 256;
        if (i < a.n) {
            bd[j] = ld(a.boards, i);
            act[j] = a.actions[i];
        }
    }
#pragma unroll
    for (int j = 0; j < B; ++j) {
        uint64_t i = base + (uint64_t)j * 256;
        if (i >= a.n) continue;
        uint32_t w0, w1, w2, w3;
        env_block(a.key, i, 0u, w0, w1, w2, w3);
        uint32_t reward;
        bool changed, done;
        transition_v1(bd[j], act[j], w0, w1, reward, changed, done);
        if ((a.flags & 1u) && done) {
            env_block(a.key, i, 1u, w0, w1, w2, w3);
            bd[j] = fresh_board(w0, w1, w2, w3);
        }
        st(a.boards, i, bd[j]);
        a.reward[i] = (int32_t)reward;
        a.done[i] = done ? 1 : 0;
    }
}

// ---- persistent grid-stride with software prefetch of the next board --------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_persist(Args a)
{
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    Board nxt = ld(a.boards, i);
    uint32_t nact = a.actions[i];
    while (true) {
        Board bd = nxt;
        uint32_t act = nact;
        const uint64_t in = i + stride;
        if (in < a.n) {
            nxt = ld(a.boards, in);
            nact = a.actions[in];
        }
        uint32_t w0, w1, w2, w3;
        env_block(a.key, i, 0u, w0, w1, w2, w3);
        uint32_t reward;
        bool changed, done;
        transition_v1(bd, act, w0, w1, reward, changed, done);
        if ((a.flags & 1u) && done) {
            env_block(a.key, i, 1u, w0, w1, w2, w3);
            bd = fresh_board(w0, w1, w2, w3);
        }
        st(a.boards, i, bd);
        a.reward[i] = (int32_t)reward;
        a.done[i] = done ? 1 : 0;
        if (in >= a.n) break;
        i = in;
    }
}

// ---- integer multiply rate probe ---------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mulrate(uint32_t *out, int iters, int mode)
{
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x12345u, z = x + 77u, w = ~x;
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) {
            x = __umulhi(x, 0xD2511F53u) ^ y;
            y = __umulhi(y, 0xCD9E8D57u) ^ z;
            z = __umulhi(z, 0xD2511F53u) ^ w;
            w = __umulhi(w, 0xCD9E8D57u) ^ x;
        }
    } else if (mode == 1) {
        for (int i = 0; i < iters; ++i) {
            x = (x * 0xD2511F53u) ^ y;
            y = (y * 0xCD9E8D57u) ^ z;
            z = (z * 0xD2511F53u) ^ w;
            w = (w * 0xCD9E8D57u) ^ x;
        }
    } else if (mode == 3) {  // one v_mad_u64_u32 per product: hi and lo from a single instruction
        for (int i = 0; i < iters; ++i) {
            uint64_t p0, p1;
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(x), "v"(0xD2511F53u) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(z), "v"(0xCD9E8D57u) : "vcc");
            x = (uint32_t)(p1 >> 32) ^ y;
            y = (uint32_t)p1;
            z = (uint32_t)(p0 >> 32) ^ w;
            w = (uint32_t)p0;
        }
    } else if (mode == 4) {  // the same Philox-like round with separate mul_hi / mul_lo
        for (int i = 0; i < iters; ++i) {
            uint32_t h0 = __umulhi(x, 0xD2511F53u), l0 = x * 0xD2511F53u;
            uint32_t h1 = __umulhi(z, 0xCD9E8D57u), l1 = z * 0xCD9E8D57u;
            x = h1 ^ y;
            y = l1;
            z = h0 ^ w;
            w = l0;
        }
    } else {
        for (int i = 0; i < iters; ++i) {
            x = (x + 0xD2511F53u) ^ y;
            y = (y + 0xCD9E8D57u) ^ z;
            z = (z + 0xD2511F53u) ^ w;
            w = (w + 0xCD9E8D57u) ^ x;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x ^ y ^ z ^ w;
}


int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int launches = argc > 2 ? atoi(argv[2]) : 300;
    const uint64_t n = 1ull << lg;
    uint4 *boards;
    uint8_t *actions, *done;
    int32_t *reward;
    // MB_SETS=k (default 1): k disjoint (boards, actions, reward, done) buffer sets, launch t works on set t % k -- with
    // k x 22 B x n beyond twice the 256 MiB Infinity Cache every launch reads and writes lines that left the cache long ago
    // (the HBM-COLD regime of bench.py's cold leg); the action pool then has max(8, k) rows.
    const int sets = getenv("MB_SETS") ? std::max(1, atoi(getenv("MB_SETS"))) : 1;
    const int act_rows = std::max(8, sets);
    CHECK(hipMalloc(&boards, n * 16 * sets));
    CHECK(hipMalloc(&actions, n * act_rows));
    CHECK(hipMalloc(&done, n * sets));
    CHECK(hipMalloc(&reward, n * 4 * sets));
    std::vector<uint8_t> h(n * act_rows);
    {
        // truly random actions: structured patterns toggle less data and let the chip clock higher (DVFS),
        // which flatters every variant by ~10 % (measured) -- bench on realistic data.
        std::mt19937 gen(1);
        for (size_t i = 0; i < h.size(); i += 4) {
            uint32_t r = gen();
            for (int k = 0; k < 4; ++k) h[i + k] = (uint8_t)((r >> (8 * k)) & 3);
        }
    }
    CHECK(hipMemcpy(actions, h.data(), h.size(), hipMemcpyHostToDevice));
    hipStream_t s = nullptr;
    if (!getenv("MB_NULL_STREAM")) CHECK(hipStreamCreate(&s));
    CHECK(env2048_reset(0, s, n, 0, (uint8_t *)boards, done, 0x2048, 0) ? hipErrorUnknown : hipSuccess);
    const int warm_steps = getenv("MB_WARM") ? atoi(getenv("MB_WARM")) : 64;
    CHECK(env2048_rollout(0, s, n, 0, (uint8_t *)boards, 0x2048, 0, warm_steps, nullptr, nullptr, nullptr, 3) ? hipErrorUnknown : hipSuccess);
    CHECK(hipStreamSynchronize(s));
    std::vector<uint4> saved(n);
    CHECK(hipMemcpy(saved.data(), boards, n * 16, hipMemcpyDeviceToHost));

    typedef int (*step_fn_t)(int, void *, uint64_t, uint64_t, uint8_t *, const uint8_t *, int32_t *, uint8_t *, uint8_t *, float *,
                             uint64_t, uint64_t, uint32_t);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // Variants are registered first, then timed in interleaved ROUNDS (guide rule 24): every round
    // runs each variant for `launches` back-to-back launches; min and median over rounds are reported.
    struct Variant {
        std::string name;
        std::function<void(int)> launch;
        std::vector<double> us;
    };
    std::vector<Variant> variants;
    auto run = [&](const char *name, std::function<void(int)> launch) { variants.push_back({name, launch, {}}); };
    auto time_all = [&](int rounds) {
        for (int r = 0; r < rounds; ++r) {
            for (auto &v : variants) {
                for (int k = 0; k < sets; ++k) CHECK(hipMemcpy(boards + (uint64_t)k * n, saved.data(), n * 16, hipMemcpyHostToDevice));
                for (int t = 0; t < 20; ++t) v.launch(t);
                CHECK(hipStreamSynchronize(s));
                CHECK(hipEventRecord(e0, s));
                for (int t = 0; t < launches; ++t) v.launch(100 + t);
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                v.us.push_back(ms * 1e3 / launches);
            }
        }
        for (auto &v : variants) {
            std::sort(v.us.begin(), v.us.end());
            double mn = v.us.front(), med = v.us[v.us.size() / 2];
            printf("%-28s min %8.2f  med %8.2f us/launch  %7.2f Gsteps/s(med)  %7.1f GB/s(38B,med)\n", v.name.c_str(), mn, med,
                   n / med * 1e-3, 38.0 * n / med * 1e-3);
        }
        fflush(stdout);
    };
    // buffers of launch t: set t % sets (one set: the action pool still rotates over its 8 rows)
    auto B = [=](int t) { return boards + (uint64_t)(t % sets) * n; };
    auto A = [=](int t) { return actions + (uint64_t)(sets > 1 ? t % sets : (t & 7)) * n; };
    auto R = [=](int t) { return reward + (uint64_t)(t % sets) * n; };
    auto D = [=](int t) { return done + (uint64_t)(t % sets) * n; };
    auto mk = [&](int t) {
        Args a{n, B(t), A(t), R(t), D(t), RngKey{0x2048u, 0u, (uint32_t)t, 0u}, 1u};
        return a;
    };
    const unsigned g256 = (unsigned)((n + 255) / 256);
    run("memonly", [&](int t) { hipLaunchKernelGGL(k_memonly, dim3(g256), dim3(256), 0, s, mk(t)); });
    run("memonly_sc1_tile4", [&](int t) { hipLaunchKernelGGL(k_memonly_sc1, dim3((unsigned)((n / 4 + 511) / 512)), dim3(512), 0, s, mk(t)); });
    run("philoxonly", [&](int t) { hipLaunchKernelGGL(k_philoxonly, dim3(g256), dim3(256), 0, s, mk(t)); });
    run("nophilox", [&](int t) { hipLaunchKernelGGL(k_nophilox, dim3(g256), dim3(256), 0, s, mk(t)); });
    run("v1_b256", [&](int t) { hipLaunchKernelGGL(k_v1<256>, dim3(g256), dim3(256), 0, s, mk(t)); });
    run("v1_b128", [&](int t) { hipLaunchKernelGGL(k_v1<128>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, mk(t)); });
    run("v1_b512", [&](int t) { hipLaunchKernelGGL(k_v1<512>, dim3((unsigned)((n + 511) / 512)), dim3(512), 0, s, mk(t)); });
    run("v1_b1024", [&](int t) { hipLaunchKernelGGL(k_v1<1024>, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, s, mk(t)); });
    run("multi2", [&](int t) { hipLaunchKernelGGL(k_multi<2>, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, s, mk(t)); });
    run("multi4", [&](int t) { hipLaunchKernelGGL(k_multi<4>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, mk(t)); });
    for (int bpc : {2, 4, 8}) {
        char name[64];
        snprintf(name, sizeof name, "persist256_x%d", bpc);
        run(name, [&](int t) { hipLaunchKernelGGL(k_persist<256>, dim3(256 * bpc), dim3(256), 0, s, mk(t)); });
    }
    run("persist512_x4", [&](int t) { hipLaunchKernelGGL(k_persist<512>, dim3(256 * 4), dim3(512), 0, s, mk(t)); });
    run("persist1024_x2", [&](int t) { hipLaunchKernelGGL(k_persist<1024>, dim3(256 * 2), dim3(1024), 0, s, mk(t)); });

    auto mk4 = [&](int t) {
        Args4 a{n, B(t), A(t), R(t), D(t), RngKey{0x2048u, 0u, (uint32_t)t, 0u}, 1u};
        return a;
    };
    const unsigned g4 = (unsigned)((n / 4 + 255) / 256);
    run("v1_ldsdir", [&](int t) { hipLaunchKernelGGL(k_v1_ldsdir<256>, dim3(g256), dim3(256), 0, s, mk4(t)); });
    run("adj4_plain", [&](int t) { hipLaunchKernelGGL((k_four<0, false>), dim3(g4), dim3(256), 0, s, mk4(t)); });
    run("adj4_shared", [&](int t) { hipLaunchKernelGGL((k_four<1, false>), dim3(g4), dim3(256), 0, s, mk4(t)); });
    run("adj4_lds", [&](int t) { hipLaunchKernelGGL((k_four<2, false>), dim3(g4), dim3(256), 0, s, mk4(t)); });
    run("adj4_shared_lds", [&](int t) { hipLaunchKernelGGL((k_four<3, false>), dim3(g4), dim3(256), 0, s, mk4(t)); });
    run("tile4_plain", [&](int t) { hipLaunchKernelGGL((k_four<0, true>), dim3(g4), dim3(256), 0, s, mk4(t)); });
    run("tile4_shared", [&](int t) { hipLaunchKernelGGL((k_four<1, true>), dim3(g4), dim3(256), 0, s, mk4(t)); });
    run("tile4_shared_lds", [&](int t) { hipLaunchKernelGGL((k_four<3, true>), dim3(g4), dim3(256), 0, s, mk4(t)); });

    typedef int (*step_fn_t_fwd)(int);
    for (int ai = 3; ai < argc; ++ai) {  // extra experiment builds of the library: time their env2048_step
        void *h = dlopen(argv[ai], RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            fprintf(stderr, "dlopen %s: %s\n", argv[ai], dlerror());
            continue;
        }
        step_fn_t fn = (step_fn_t)dlsym(h, "env2048_step");
        std::string name = std::string("EXP ") + argv[ai];
        typedef const char *(*err_fn_t)(void);
        err_fn_t err = (err_fn_t)dlsym(h, "env2048_last_error");
        typedef int (*roll_fn_t)(int, void *, uint64_t, uint64_t, uint8_t *, uint64_t, uint64_t, uint32_t, int64_t *, uint32_t *,
                                 uint32_t *, uint32_t);
        roll_fn_t roll = (roll_fn_t)dlsym(h, "env2048_rollout");
        std::string rname = std::string("EXPROLL16 ") + argv[ai];
        run(rname.c_str(), [=](int t) {
            int rc = roll(0, s, n, 0, (uint8_t *)B(t), 0x2048, 16 * (uint64_t)t, 16, nullptr, nullptr, nullptr, 3);
            if (rc) {
                fprintf(stderr, "env2048_rollout failed: %d %s\n", rc, err());
                exit(2);
            }
        });
        typedef int (*many_fn_t)(int, void *, uint64_t, const env2048_step_args *);
        many_fn_t many = (many_fn_t)dlsym(h, "env2048_step_many");
        std::string mname = std::string("EXP step_many ") + argv[ai];
        if (many && sets > 1)
            run(mname.c_str(), [=](int t) {  // the experiment build's shared launch over ALL sets, once every `sets` launches
                if (t % sets) return;
                std::vector<env2048_step_args> recs(sets);
                for (int k = 0; k < sets; ++k)
                    recs[k] = env2048_step_args{n, (uint64_t)k * n, (uint8_t *)B(k), A(k), R(k), D(k), nullptr, nullptr, 0x2048, (uint64_t)t, 1, 0, nullptr, nullptr};
                if (many(0, s, sets, recs.data())) {
                    fprintf(stderr, "env2048_step_many failed: %s\n", err());
                    exit(2);
                }
            });
        run(name.c_str(), [=](int t) {
            static const uint32_t exp_flags = getenv("MB_FLAGS") ? (uint32_t)atoi(getenv("MB_FLAGS")) : 1u;  // e.g. 33 = AUTO_RESET | STORES_WRITE_BACK
            int rc = fn(0, s, n, 0, (uint8_t *)B(t), A(t), R(t), D(t), nullptr, nullptr, 0x2048, t, exp_flags);
            if (rc) {
                fprintf(stderr, "env2048_step failed: %d %s\n", rc, err());
                exit(2);
            }
        });
    }
    run("LIB env2048_step", [&](int t) {
        env2048_step(0, s, n, 0, (uint8_t *)B(t), A(t), R(t), D(t), nullptr, nullptr, 0x2048, t, 1);
    });
    static const uint32_t many_flags = getenv("MB_MANY_FLAGS") ? (uint32_t)atoi(getenv("MB_MANY_FLAGS")) : 1u;  // record flags of the shared launch
    run("LIB step_many x sets (per launch)", [&](int t) {  // the fused multi-record launch over ALL sets, once every `sets` launches
        if (t % sets) return;
        static std::vector<env2048_step_args> recs;
        recs.resize(sets);
        for (int k = 0; k < sets; ++k)
            recs[k] = env2048_step_args{n, (uint64_t)k * n, (uint8_t *)B(k), A(k), R(k), D(k), nullptr, nullptr, 0x2048, (uint64_t)t, many_flags, 0, nullptr, nullptr};
        env2048_step_many(0, s, sets, recs.data());
    });
    run("LIB step no-reset", [&](int t) {
        env2048_step(0, s, n, 0, (uint8_t *)boards, actions + (uint64_t)(t & 7) * n, reward, done, nullptr, nullptr, 0x2048, t, 0);
    });
    run("LIB step random", [&](int t) {
        env2048_step(0, s, n, 0, (uint8_t *)boards, nullptr, reward, done, nullptr, nullptr, 0x2048, t, 3);
    });
    {
        static uint8_t *legal_buf = nullptr;
        static float *obs_buf = nullptr;
        CHECK(hipMalloc(&legal_buf, n));
        CHECK(hipMalloc(&obs_buf, n * 64));
        uint8_t *lb = legal_buf;
        float *ob = obs_buf;
        run("LIB step +legal", [=](int t) {
            env2048_step(0, s, n, 0, (uint8_t *)boards, actions + (uint64_t)(t & 7) * n, reward, done, lb, nullptr, 0x2048, t, 1);
        });
        run("LIB step +legal+obs", [=](int t) {
            env2048_step(0, s, n, 0, (uint8_t *)boards, actions + (uint64_t)(t & 7) * n, reward, done, lb, ob, 0x2048, t, 1);
        });
        run("LIB status legal+obs", [=](int t) {
            env2048_status(0, s, n, (uint8_t *)boards, nullptr, lb, ob, nullptr, nullptr);
        });
    }
    run("LIB rollout T=16 (/16)", [&](int t) {
        env2048_rollout(0, s, n, 0, (uint8_t *)boards, 0x2048, 16 * t, 16, nullptr, nullptr, nullptr, 3);
    });

    if (getenv("MB_STAMPS")) {  // wave timeline from the diagnostic build (tools/libexp_stamps.so)
        void *h = dlopen(getenv("MB_STAMPS"), RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
        step_fn_t fn = (step_fn_t)dlsym(h, "env2048_step");
        const uint64_t n_waves = n / 256;
        uint64_t *stamps;
        CHECK(hipMalloc(&stamps, n_waves * 64));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int t = 0; t < 200; ++t) {
            if (t == 199) CHECK(hipEventRecord(e0, s));
            fn(0, s, n, 0, (uint8_t *)boards, actions + (uint64_t)(t & 7) * n, reward, done, nullptr, (float *)stamps, 0x2048, t, 1);
        }
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float last_ms = 0;
        CHECK(hipEventElapsedTime(&last_ms, e0, e1));
        std::vector<uint64_t> hst(n_waves * 8);
        CHECK(hipMemcpy(hst.data(), stamps, n_waves * 64, hipMemcpyDeviceToHost));
        // all stamps are s_memrealtime ticks (100 MHz, chip-global): 0.01 us each
        uint64_t t0 = ~0ull;
        for (uint64_t w = 0; w < n_waves; ++w) t0 = std::min(t0, hst[8 * w]);
        const char *names[5] = {"wave start", "half-1 home", "half-1 done", "half-2 home", "wave end"};
        auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
        printf("last launch by events: %.2f us; stamps in us since the first wave start\n", last_ms * 1e3);
        for (int k = 0; k < 5; ++k) {
            std::vector<double> v;
            for (uint64_t w = 0; w < n_waves; ++w) v.push_back((double)(hst[8 * w + k] - t0) / 100.0);
            printf("  %-12s p0 %6.2f  p10 %6.2f  p25 %6.2f  p50 %6.2f  p75 %6.2f  p90 %6.2f  p100 %6.2f\n", names[k], pct(v, 0), pct(v, .1),
                   pct(v, .25), pct(v, .5), pct(v, .75), pct(v, .9), pct(v, 1));
        }
        const char *dnames[4] = {"start->h1home", "h1 compute", "wait h2", "h2+stores"};
        for (int k = 0; k < 4; ++k) {
            std::vector<double> v;
            for (uint64_t w = 0; w < n_waves; ++w) v.push_back((double)(hst[8 * w + k + 1] - hst[8 * w + k]) / 100.0);
            printf("  %-12s p0 %6.2f  p10 %6.2f  p25 %6.2f  p50 %6.2f  p75 %6.2f  p90 %6.2f  p100 %6.2f\n", dnames[k], pct(v, 0), pct(v, .1),
                   pct(v, .25), pct(v, .5), pct(v, .75), pct(v, .9), pct(v, 1));
        }
        if (getenv("MB_STAMPS_DUMP")) {  // one SIMD's waves: hw_id fields
            FILE *f = fopen(getenv("MB_STAMPS_DUMP"), "w");
            for (uint64_t w = 0; w < n_waves; ++w)
                fprintf(f, "%llu %llu %llu %llu %llu %llu 0x%llx\n", (unsigned long long)w, (unsigned long long)(hst[8 * w] - t0),
                        (unsigned long long)(hst[8 * w + 1] - t0), (unsigned long long)(hst[8 * w + 2] - t0),
                        (unsigned long long)(hst[8 * w + 3] - t0), (unsigned long long)(hst[8 * w + 4] - t0),
                        (unsigned long long)hst[8 * w + 5]);
            fclose(f);
        }
        return 0;
    }

    // Multi-stream: the batch cut into S contiguous slices, slice k always on stream k (no cross-stream
    // dependency between consecutive steps); timed with the host clock around a full sync.
    for (int S : {1, 2, 4, 8}) {
        if (getenv("MB_ONLY")) break;  // (a filtered run wants its variants only)
        std::vector<hipStream_t> ss(S);
        for (auto &x : ss) CHECK(hipStreamCreate(&x));
        const uint64_t per = n / S;
        CHECK(hipMemcpy(boards, saved.data(), n * 16, hipMemcpyHostToDevice));
        auto go = [&](int t) {
            for (int k = 0; k < S; ++k)
                env2048_step(0, ss[k], per, k * per, (uint8_t *)boards + 16 * k * per, actions + (uint64_t)(t & 7) * n + k * per,
                             reward + k * per, done + k * per, nullptr, nullptr, 0x2048, t, 1);
        };
        double best = 1e30;
        for (int r = 0; r < 5; ++r) {
            for (int t = 0; t < 50; ++t) go(t);
            CHECK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int t = 0; t < launches; ++t) go(100 + t);
            CHECK(hipDeviceSynchronize());
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
            best = std::min(best, us);
        }
        printf("STREAMS %d x %llu boards: %8.2f us/step  %7.2f Gsteps/s  %7.1f GB/s(38B)\n", S, (unsigned long long)per, best,
               n / best * 1e-3, 38.0 * n / best * 1e-3);
        for (auto &x : ss) CHECK(hipStreamDestroy(x));
    }

    const char *only = getenv("MB_ONLY");  // comma-free substring filter, e.g. MB_ONLY=EXP
    if (only) {
        std::vector<Variant> kept;
        for (auto &v : variants)
            if (v.name.find(only) != std::string::npos || v.name.find("memonly") == 0) kept.push_back(v);
        variants.swap(kept);
    }
    time_all(getenv("MB_ROUNDS") ? atoi(getenv("MB_ROUNDS")) : 7);
    if (only) return 0;

    // multiply-rate probe: 256 CUs x 8 blocks, 4096 iterations x 4 dependent chains
    uint32_t *out;
    CHECK(hipMalloc(&out, 2048 * 256 * 4));
    for (int mode = 0; mode < 5; ++mode) {
        const int iters = 4096;
        hipLaunchKernelGGL(k_mulrate, dim3(2048), dim3(256), 0, s, out, iters, mode);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_mulrate, dim3(2048), dim3(256), 0, s, out, iters, mode);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // wave-instructions per SIMD: 2048 blocks*4 waves / 1024 SIMDs = 8 waves/SIMD, each iters*8 instr (mul+xor)
        double instr_per_simd = 8.0 * iters * 8;
        printf("mulrate mode %d (%s): %.3f ms -> %.2f ns per wave-instr-pair-slot, ~%.2f cycles/instr at 2.4GHz\n", mode,
               mode == 0 ? "mul_hi+xor" : mode == 1 ? "mul_lo+xor" : mode == 2 ? "add+xor" : mode == 3 ? "philox round via 2x mad_u64_u32 (4 instr/iter)" : "philox round via 2x(mul_hi+mul_lo) (6 instr/iter)", ms, ms * 1e6 / instr_per_simd,
               ms * 1e6 / instr_per_simd * 2.4);
    }
    return 0;
}
