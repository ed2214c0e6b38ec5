// env2048_kernels.hip -- HIP kernels (gfx950) and the C-ABI of include/env2048.h.
//
// Data layout in HBM: `boards` is an array of 16-byte records (one global_load_dwordx4 /
// global_store_dwordx4 per board, a wave moves 1 KiB contiguous); actions/done/legal are byte
// arrays, reward int32, obs 16 floats per board.  One board per lane, all arithmetic in VGPRs
// (board_ops.hpp); no LDS, no MFMA -- this path is integer/indexing work bounded by HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "env2048.h"
#include "board_ops.hpp"

using namespace env2048;

namespace {

constexpr int BLOCK = 256;

__device__ __forceinline__ Board load_board(const uint8_t *boards, uint64_t i)
{
    uint4 v = reinterpret_cast<const uint4 *>(boards)[i];
    return Board{v.x, v.y, v.z, v.w};
}

__device__ __forceinline__ void store_board(uint8_t *boards, uint64_t i, const Board &bd)
{
    reinterpret_cast<uint4 *>(boards)[i] = make_uint4(bd.r0, bd.r1, bd.r2, bd.r3);
}

// log2(tile)/16 per cell [ref: reinforce/game/core.py:346-377 encode_observation]; exact in f32.
__device__ __forceinline__ void store_obs(float *obs, uint64_t i, const Board &bd)
{
    float4 *o = reinterpret_cast<float4 *>(obs) + 4 * i;
    const uint32_t rows[4] = {bd.r0, bd.r1, bd.r2, bd.r3};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t x = rows[r];
        o[r] = make_float4((float)(x & 0xFFu) * 0.0625f, (float)((x >> 8) & 0xFFu) * 0.0625f,
                           (float)((x >> 16) & 0xFFu) * 0.0625f, (float)(x >> 24) * 0.0625f);
    }
}

struct StepArgs {
    uint64_t n;
    uint64_t env_base;
    uint8_t *boards;
    const uint8_t *actions;
    int32_t *reward;
    uint8_t *done;
    uint8_t *legal;
    float *obs;
    RngKey key;
    uint32_t flags;
};

// EXTRAS: legal and/or obs requested (kept out of the base variant so it stays lean).
template <bool EXTRAS>
__global__ __launch_bounds__(BLOCK) void step_kernel(StepArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const uint64_t env = a.env_base + i;
    Board bd = load_board(a.boards, i);
    uint32_t w0, w1, w2, w3;
    env_block(a.key, env, 0u, w0, w1, w2, w3);
    const uint32_t action = (a.flags & ENV2048_FLAG_RANDOM_ACTIONS) ? w2 : (uint32_t)a.actions[i];
    uint32_t reward;
    bool changed, done;
    transition(bd, action, w0, w1, reward, changed, done);
    if ((a.flags & ENV2048_FLAG_AUTO_RESET) && done) {
        env_block(a.key, env, 1u, w0, w1, w2, w3);
        bd = fresh_board(w0, w1, w2, w3);
    }
    store_board(a.boards, i, bd);
    if (a.reward) a.reward[i] = (int32_t)reward;
    a.done[i] = done ? 1 : 0;
    if (EXTRAS) {
        if (a.legal) a.legal[i] = (uint8_t)legal_mask(bd);
        if (a.obs) store_obs(a.obs, i, bd);
    }
}

struct ResetArgs {
    uint64_t n;
    uint64_t env_base;
    uint8_t *boards;
    uint8_t *done;
    RngKey key;
};

__global__ __launch_bounds__(BLOCK) void reset_kernel(ResetArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    uint32_t w0, w1, w2, w3;
    env_block(a.key, a.env_base + i, 1u, w0, w1, w2, w3);
    store_board(a.boards, i, fresh_board(w0, w1, w2, w3));
    if (a.done) a.done[i] = 0;
}

struct RolloutArgs {
    uint64_t n;
    uint64_t env_base;
    uint8_t *boards;
    int64_t *return_sum;
    uint32_t *episodes;
    uint32_t *valid_moves;
    uint32_t seed_lo, seed_hi;
    uint64_t step0;
    uint32_t T;
    uint32_t flags;
};

__global__ __launch_bounds__(BLOCK) void rollout_kernel(RolloutArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const uint64_t env = a.env_base + i;
    Board bd = load_board(a.boards, i);
    int64_t ret = 0;
    uint32_t eps = 0, moves = 0;
    for (uint32_t t = 0; t < a.T; ++t) {
        const uint64_t step = a.step0 + t;
        const RngKey key{a.seed_lo, a.seed_hi, (uint32_t)step, (uint32_t)(step >> 32)};
        uint32_t w0, w1, w2, w3;
        env_block(key, env, 0u, w0, w1, w2, w3);
        uint32_t reward;
        bool changed, done;
        transition(bd, w2, w0, w1, reward, changed, done);
        if ((a.flags & ENV2048_FLAG_AUTO_RESET) && done) {
            env_block(key, env, 1u, w0, w1, w2, w3);
            bd = fresh_board(w0, w1, w2, w3);
        }
        ret += reward;
        eps += done ? 1u : 0u;
        moves += changed ? 1u : 0u;
    }
    store_board(a.boards, i, bd);
    if (a.return_sum) a.return_sum[i] = ret;
    if (a.episodes) a.episodes[i] = eps;
    if (a.valid_moves) a.valid_moves[i] = moves;
}

__global__ __launch_bounds__(BLOCK) void afterstate_kernel(uint64_t n, const uint8_t *boards_in,
                                                           const uint8_t *actions, uint8_t *boards_out,
                                                           int32_t *reward, uint8_t *changed_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Board bd = load_board(boards_in, i);
    uint32_t score;
    bool changed;
    afterstate(bd, (uint32_t)actions[i] & 3u, score, changed);
    store_board(boards_out, i, bd);
    if (reward) reward[i] = (int32_t)score;
    if (changed_out) changed_out[i] = changed ? 1 : 0;
}

__global__ __launch_bounds__(BLOCK) void spawn_kernel(uint64_t n, uint8_t *boards, const uint8_t *spawn_k,
                                                      const uint8_t *spawn_e, uint8_t *placed)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Board bd = load_board(boards, i);
    const uint32_t empt = empty_mask(bd);
    const uint32_t k = spawn_k[i];
    const bool ok = k < (uint32_t)__popc(empt);
    const uint32_t bit = ok ? kth_empty(empt, k) : 0u;
    const uint32_t e = spawn_e[i];
    // general exponent (not only 1/2): OR the byte value into the selected (empty) cell
    bd.r0 |= ((bit >> 7) & K01) * e;
    bd.r1 |= ((bit >> 6) & K01) * e;
    bd.r2 |= ((bit >> 5) & K01) * e;
    bd.r3 |= ((bit >> 4) & K01) * e;
    store_board(boards, i, bd);
    if (placed) placed[i] = ok ? (uint8_t)cell_index(bit) : (uint8_t)0xFF;
}

__global__ __launch_bounds__(BLOCK) void status_kernel(uint64_t n, const uint8_t *boards, uint8_t *done,
                                                       uint8_t *legal, float *obs, uint8_t *max_exp,
                                                       uint8_t *n_empty)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Board bd = load_board(boards, i);
    const uint32_t empties = count_empty(bd);
    if (done) done[i] = (empties == 0u && no_equal_neighbours(bd)) ? 1 : 0;
    if (legal) legal[i] = (uint8_t)legal_mask(bd);
    if (obs) store_obs(obs, i, bd);
    if (n_empty) n_empty[i] = (uint8_t)empties;
    if (max_exp) {
        const uint32_t rows[4] = {bd.r0, bd.r1, bd.r2, bd.r3};
        uint32_t m = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            m = max(m, rows[r] & 0xFFu);
            m = max(m, (rows[r] >> 8) & 0xFFu);
            m = max(m, (rows[r] >> 16) & 0xFFu);
            m = max(m, rows[r] >> 24);
        }
        max_exp[i] = (uint8_t)m;
    }
}

// values <-> exponents; one thread per 4 cells (16-byte value loads for int32, coalesced).
template <typename T>
__global__ __launch_bounds__(BLOCK) void pack_kernel(uint64_t n_quads, const T *values, uint32_t *packed)
{
    const uint64_t q = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= n_quads) return;
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t v = (uint64_t)values[4 * q + j];
        const uint32_t e = v ? (uint32_t)(63 - __clzll((long long)v)) : 0u;
        out |= (e & 0xFFu) << (8 * j);
    }
    packed[q] = out;
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void unpack_kernel(uint64_t n_quads, const uint32_t *packed, T *values)
{
    const uint64_t q = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= n_quads) return;
    const uint32_t x = packed[q];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t e = (x >> (8 * j)) & 0xFFu;
        values[4 * q + j] = e ? (T)((uint64_t)1 << (e & 63u)) : (T)0;
    }
}

// ------------------------------------------------------------------------------------------
// Host side of the C-ABI
// ------------------------------------------------------------------------------------------
thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    return fail(ENV2048_ERR_HIP_BASE - (int)e, "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
}

bool misaligned(const void *p, uintptr_t a) { return p && (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; }

// Select `device` for the duration of one entry point and put the caller's device back after.
struct DeviceScope {
    int prev = -1;
    int rc = ENV2048_OK;
    explicit DeviceScope(int device)
    {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0) {
            rc = fail(ENV2048_ERR_DEVICE, "no usable HIP device (%s); this library has no CPU fallback",
                      e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
            return;
        }
        if (device < 0 || device >= count) {
            rc = fail(ENV2048_ERR_DEVICE, "device %d out of range [0, %d)", device, count);
            return;
        }
        e = hipGetDevice(&prev);
        if (e != hipSuccess) {
            rc = hip_fail(e, "hipGetDevice");
            prev = -1;
            return;
        }
        if (prev != device) {
            e = hipSetDevice(device);
            if (e != hipSuccess) {
                rc = hip_fail(e, "hipSetDevice");
                prev = -1;
            }
        } else {
            prev = -1;
        }
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int check_range(uint64_t n, uint64_t env_base)
{
    // env_hi is folded into counter word 3 as 2*env_hi + stream: keep env below 2^63.
    if (env_base + n < env_base || (env_base + n) >> 63)
        return fail(ENV2048_ERR_RANGE, "env_base + n = %llu + %llu exceeds the 2^63 environment index range",
                    (unsigned long long)env_base, (unsigned long long)n);
    return ENV2048_OK;
}

int check_count(uint64_t n)
{
    if (n > ((uint64_t)0x7FFFFFFF) * BLOCK)
        return fail(ENV2048_ERR_RANGE, "n = %llu exceeds the largest single launch (%llu boards)",
                    (unsigned long long)n, (unsigned long long)0x7FFFFFFF * BLOCK);
    return ENV2048_OK;
}

inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }

int launched(const char *what)
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ENV2048_OK : hip_fail(e, what);
}

}  // namespace

extern "C" {

int env2048_abi_version(void) { return ENV2048_ABI_VERSION; }

const char *env2048_last_error(void) { return g_error; }

int env2048_device_count(void)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return fail(ENV2048_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return count;
}

int env2048_reset(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, uint8_t *done,
                  uint64_t seed, uint64_t step_index)
{
    if (n == 0) return ENV2048_OK;
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_reset: boards is NULL");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_reset: boards must be 16-byte aligned");
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    ResetArgs a{n, env_base, boards, done,
                RngKey{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step_index, (uint32_t)(step_index >> 32)}};
    hipLaunchKernelGGL(reset_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return launched("env2048_reset launch");
}

int env2048_step(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards,
                 const uint8_t *actions, int32_t *reward, uint8_t *done, uint8_t *legal, float *obs,
                 uint64_t seed, uint64_t step_index, uint32_t flags)
{
    if (n == 0) return ENV2048_OK;
    if (flags & ~(ENV2048_FLAG_AUTO_RESET | ENV2048_FLAG_RANDOM_ACTIONS))
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_step: unknown flag bits 0x%x", flags);
    if (!boards || !done) return fail(ENV2048_ERR_INVALID_ARG, "env2048_step: boards and done are required");
    if (!actions && !(flags & ENV2048_FLAG_RANDOM_ACTIONS))
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_step: actions is NULL without ENV2048_FLAG_RANDOM_ACTIONS");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_step: boards must be 16-byte aligned");
    if (misaligned(reward, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_step: reward must be 4-byte aligned");
    if (misaligned(obs, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_step: obs must be 16-byte aligned");
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    StepArgs a{n, env_base, boards, actions, reward, done, legal, obs,
               RngKey{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step_index, (uint32_t)(step_index >> 32)},
               flags};
    if (legal || obs)
        hipLaunchKernelGGL(step_kernel<true>, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(step_kernel<false>, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return launched("env2048_step launch");
}

int env2048_rollout(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, uint64_t seed,
                    uint64_t step0, uint32_t T, int64_t *return_sum, uint32_t *episodes, uint32_t *valid_moves,
                    uint32_t flags)
{
    if (n == 0) return ENV2048_OK;
    if (flags & ~(ENV2048_FLAG_AUTO_RESET | ENV2048_FLAG_RANDOM_ACTIONS))
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_rollout: unknown flag bits 0x%x", flags);
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_rollout: boards is NULL");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_rollout: boards must be 16-byte aligned");
    if (misaligned(return_sum, 8) || misaligned(episodes, 4) || misaligned(valid_moves, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_rollout: misaligned output array");
    if (step0 + T < step0) return fail(ENV2048_ERR_RANGE, "env2048_rollout: step0 + T overflows");
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    RolloutArgs a{n, env_base, boards, return_sum, episodes, valid_moves, (uint32_t)seed, (uint32_t)(seed >> 32),
                  step0, T, flags};
    hipLaunchKernelGGL(rollout_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return launched("env2048_rollout launch");
}

int env2048_afterstate(int device, void *stream, uint64_t n, const uint8_t *boards_in, const uint8_t *actions,
                       uint8_t *boards_out, int32_t *reward, uint8_t *changed)
{
    if (n == 0) return ENV2048_OK;
    if (!boards_in || !actions || !boards_out)
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_afterstate: boards_in, actions and boards_out are required");
    if (misaligned(boards_in, 16) || misaligned(boards_out, 16))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_afterstate: boards must be 16-byte aligned");
    if (misaligned(reward, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_afterstate: reward must be 4-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(afterstate_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards_in, actions,
                       boards_out, reward, changed);
    return launched("env2048_afterstate launch");
}

int env2048_spawn(int device, void *stream, uint64_t n, uint8_t *boards, const uint8_t *spawn_k,
                  const uint8_t *spawn_e, uint8_t *placed)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !spawn_k || !spawn_e)
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_spawn: boards, spawn_k and spawn_e are required");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_spawn: boards must be 16-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(spawn_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, spawn_k, spawn_e,
                       placed);
    return launched("env2048_spawn launch");
}

int env2048_status(int device, void *stream, uint64_t n, const uint8_t *boards, uint8_t *done, uint8_t *legal,
                   float *obs, uint8_t *max_exp, uint8_t *n_empty)
{
    if (n == 0) return ENV2048_OK;
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_status: boards is NULL");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_status: boards must be 16-byte aligned");
    if (misaligned(obs, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_status: obs must be 16-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(status_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, done, legal, obs,
                       max_exp, n_empty);
    return launched("env2048_status launch");
}

#define ENV2048_DEFINE_PACK(SUFFIX, TYPE)                                                                          \
    int env2048_pack_##SUFFIX(int device, void *stream, uint64_t n, const TYPE *values, uint8_t *boards)           \
    {                                                                                                              \
        if (n == 0) return ENV2048_OK;                                                                             \
        if (!values || !boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_pack_" #SUFFIX ": NULL argument");   \
        if (misaligned(boards, 16) || misaligned(values, sizeof(TYPE)))                                            \
            return fail(ENV2048_ERR_ALIGNMENT, "env2048_pack_" #SUFFIX ": misaligned buffer");                     \
        if (int rc = check_count(4 * n)) return rc;                                                                \
        DeviceScope scope(device);                                                                                 \
        if (scope.rc) return scope.rc;                                                                             \
        hipLaunchKernelGGL(pack_kernel<TYPE>, grid_for(4 * n), dim3(BLOCK), 0, (hipStream_t)stream, 4 * n, values, \
                           reinterpret_cast<uint32_t *>(boards));                                                  \
        return launched("env2048_pack_" #SUFFIX " launch");                                                        \
    }                                                                                                              \
    int env2048_unpack_##SUFFIX(int device, void *stream, uint64_t n, const uint8_t *boards, TYPE *values)         \
    {                                                                                                              \
        if (n == 0) return ENV2048_OK;                                                                             \
        if (!values || !boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_unpack_" #SUFFIX ": NULL argument"); \
        if (misaligned(boards, 16) || misaligned(values, sizeof(TYPE)))                                            \
            return fail(ENV2048_ERR_ALIGNMENT, "env2048_unpack_" #SUFFIX ": misaligned buffer");                   \
        if (int rc = check_count(4 * n)) return rc;                                                                \
        DeviceScope scope(device);                                                                                 \
        if (scope.rc) return scope.rc;                                                                             \
        hipLaunchKernelGGL(unpack_kernel<TYPE>, grid_for(4 * n), dim3(BLOCK), 0, (hipStream_t)stream, 4 * n,       \
                           reinterpret_cast<const uint32_t *>(boards), values);                                    \
        return launched("env2048_unpack_" #SUFFIX " launch");                                                      \
    }

ENV2048_DEFINE_PACK(i32, int32_t)
ENV2048_DEFINE_PACK(i64, int64_t)

}  // extern "C"
