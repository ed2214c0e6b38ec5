// transform_kernels.hip -- format and status kernels around the step path (one-hot observation, status bytes, value <->
// exponent packing, materialised random actions) plus the library-level entry points (ABI version, last error, device count).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "env2048.h"
#include "board_ops.hpp"
#include "board_io.hpp"
#include "tiling.hpp"
#include "device_scope.hpp"

using namespace env2048;
using namespace env2048::host;

namespace {
// The actions ENV2048_FLAG_RANDOM_ACTIONS makes a step take, materialised: action of environment env at (seed, step) = low two
// bits of its step word.  For recorders: a step played with these actions read from memory is the same step (same words for
// the spawn), and the action row then says what was played (recorded random steps had all-zero action rows).
// One lane per STEP BLOCK: the four environments 4 b ... 4 b + 3 share block b of the step stream (RNG contract), so a lane draws
// one Philox block and writes their four moves as one dword (bytes at the ragged ends of the batch, or when the destination
// is not dword-aligned for that lane).
__global__ __launch_bounds__(BLOCK) void random_actions_kernel(uint64_t n, uint64_t env_base, uint32_t seed_lo, uint32_t seed_hi,
                                                               uint64_t step_index, uint8_t *actions)
{
    const uint64_t block = (env_base >> 2) + (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t first = 4u * block, end = env_base + n;  // environments [first, first + 4) of the global index space
    if (first >= end) return;
    const RngKey key{seed_lo, seed_hi, (uint32_t)step_index, (uint32_t)(step_index >> 32)};
    uint32_t w[4];
    env_block(key, block, STREAM_STEP, w[0], w[1], w[2], w[3]);
    const uint32_t moves = (w[0] & 3u) | ((w[1] & 3u) << 8) | ((w[2] & 3u) << 16) | ((w[3] & 3u) << 24);
    if (first >= env_base && first + 4u <= end && (((uintptr_t)actions + (first - env_base)) & 3u) == 0u) {
        *reinterpret_cast<uint32_t *>(actions + (first - env_base)) = moves;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint64_t env = first + k;
            if (env >= env_base && env < end) actions[env - env_base] = (uint8_t)(moves >> (8u * k));
        }
    }
}

// One block per 64 boards; a thread writes whole output DWORDs (coalesced 4-byte stores): byte g of a
// board's 16*width-byte record belongs to cell g / width and is 1 iff g % width equals that cell's
// exponent.  All index arithmetic is 32-bit on block-local offsets (16 * width is a multiple of 4).
__global__ __launch_bounds__(BLOCK) void onehot_kernel(uint64_t n, const uint8_t *boards, uint32_t width, uint32_t *out)
{
    __shared__ uint8_t cells[64 * 16];
    const uint64_t board0 = (uint64_t)blockIdx.x * 64;
    const uint32_t n_boards = (uint32_t)min((uint64_t)64, n - board0);
    for (uint32_t c = threadIdx.x; c < n_boards * 16u; c += BLOCK) cells[c] = boards[board0 * 16 + c];
    __syncthreads();
    const uint32_t dwords = n_boards * 4u * width;  // 16 * width bytes per board
    uint32_t *dst = out + board0 * 4u * width;
    for (uint32_t d = threadIdx.x; d < dwords; d += BLOCK) {
        const uint32_t g0 = 4u * d;
        uint32_t cell = g0 / width;
        uint32_t pos = g0 - cell * width;
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            word |= (pos == cells[cell & 1023u] ? 1u : 0u) << (8 * b);
            if (++pos == width) {
                pos = 0;
                ++cell;
            }
        }
        dst[d] = word;
    }
}

// Same output, 16 bytes per thread (1 KiB per wave store) for 16-byte aligned `out`: instead of testing every
// output byte, a thread walks the (at most 16/width + 2) cells that overlap its chunk and drops each cell's
// single 1 into place.  One division per chunk, done as a multiply (exact for block-local offsets < 2^18).
// STREAM: non-temporal stores (device_scope.hpp stream_output()).
template <bool STREAM>
__global__ __launch_bounds__(BLOCK) void onehot_kernel16(uint64_t n, const uint8_t *boards, uint32_t width, uint32_t inv_width,
                                                         uint4 *out)
{
    __shared__ uint8_t cells[64 * 16];
    const uint64_t board0 = (uint64_t)blockIdx.x * 64;
    const uint32_t n_boards = (uint32_t)min((uint64_t)64, n - board0);
    for (uint32_t c = threadIdx.x; c < n_boards * 16u; c += BLOCK) cells[c] = boards[board0 * 16 + c];
    __syncthreads();
    const uint32_t chunks = n_boards * width;  // 16 * width bytes per board = width chunks of 16 bytes
    uint4 *dst = out + board0 * width;
    for (uint32_t q = threadIdx.x; q < chunks; q += BLOCK) {
        const uint32_t g0 = 16u * q;
        uint32_t cell = __umulhi(g0, inv_width);  // g0 / width
        uint32_t base = cell * width;             // first output byte of that cell
        uint64_t lo = 0, hi = 0;
        while (base < g0 + 16u) {
            const uint32_t e = cells[cell & 1023u];
            const uint32_t off = base + e - g0;  // position of the cell's 1 inside this chunk (wraps if before it)
            if (e < width && off < 16u) {
                if (off < 8u)
                    lo |= 1ull << (8u * off);
                else
                    hi |= 1ull << (8u * (off - 8u));
            }
            ++cell;
            base += width;
        }
        if (STREAM) {
            typedef uint32_t vec4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(vec4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)}, reinterpret_cast<vec4 *>(dst) + q);
        } else {
            store_x4_through(dst + q, u32x4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)});  // (write-through, like every wide output)
        }
    }
}

// obs_through: write-through stores for the observation rows (64 B per board), the form the step kernel uses for its own:
// measured 14.3 -> 10.7 us per 2^20 boards, 59.5 -> 58.6 at 2^22 (never slower), so the entry point always asks for it.
__global__ __launch_bounds__(256) void status_kernel(uint64_t n, const uint8_t *boards, uint8_t *done,
                                                     uint8_t *legal, float *obs, uint8_t *max_exp,
                                                     uint8_t *n_empty, bool obs_through)
{
    __shared__ __attribute__((aligned(16))) uint32_t strip_lds[4][256];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool valid = i < n;
    const Board bd = valid ? load_board(boards, i) : Board{0u, 0u, 0u, 0u};
    if (obs) store_obs_coalesced(obs, strip_lds[wave], i - lane, n, lane, bd, !valid, obs_through);  // whole wave takes part
    if (!valid) return;
    const uint32_t empties = count_empty(bd);
    if (done) done[i] = (empties == 0u && no_equal_neighbours(bd)) ? 1 : 0;
    if (legal) legal[i] = (uint8_t)legal_mask(bd);
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
        values[4 * q + j] = e ? (T)((uint64_t)1 << (e & 63u)) : (T)0;  // (plain stores: non-temporal ones halve the rate of the int64 form, measured)
    }
}

}  // namespace

extern "C" {

int env2048_abi_version(void) { return ENV2048_ABI_VERSION; }

const char *env2048_last_error(void) { return g_error; }

// A completion word for ANY work on a stream: hipStreamWriteValue64 makes the device store `value` into the caller's pinned host
// word once everything enqueued on `stream` before this call has completed; env2048_wait_sequence polls it.
int env2048_stream_signal(int device, void *stream, uint64_t *host_word, uint64_t value)
{
    if (!host_word || misaligned(host_word, 8)) return fail(ENV2048_ERR_INVALID_ARG, "env2048_stream_signal: host_word is NULL or misaligned");
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    const hipError_t e = hipStreamWriteValue64((hipStream_t)stream, host_word, value, 0);
    return e == hipSuccess ? ENV2048_OK : hip_fail(e, "env2048_stream_signal: hipStreamWriteValue64");
}

int env2048_device_count(void)
{
    hipError_t e = hipSuccess;
    const int count = visible_devices(&e);
    if (e != hipSuccess) return fail(ENV2048_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return count;
}

int env2048_random_actions(int device, void *stream, uint64_t n, uint64_t env_base, uint64_t seed, uint64_t step_index, uint8_t *actions)
{
    if (n == 0) return ENV2048_OK;
    if (!actions) return fail(ENV2048_ERR_INVALID_ARG, "env2048_random_actions: actions is NULL");
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    const uint64_t step_blocks = ((env_base + n + 3u) >> 2) - (env_base >> 2);  // step blocks the batch touches
    hipLaunchKernelGGL(random_actions_kernel, grid_for(step_blocks), dim3(BLOCK), 0, (hipStream_t)stream, n, env_base, (uint32_t)seed,
                       (uint32_t)(seed >> 32), step_index, actions);
    return launched("env2048_random_actions launch");
}

int env2048_encode_onehot(int device, void *stream, uint64_t n, const uint8_t *boards, uint32_t width, uint8_t *out)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !out) return fail(ENV2048_ERR_INVALID_ARG, "env2048_encode_onehot: NULL argument");
    if (width == 0 || width > 255) return fail(ENV2048_ERR_INVALID_ARG, "env2048_encode_onehot: width %u not in [1, 255]", width);
    if (misaligned(out, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_encode_onehot: out must be 4-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    if (!misaligned(out, 16) && stream_output(n * 16u * width))
        hipLaunchKernelGGL(onehot_kernel16<true>, dim3((unsigned)((n + 63) / 64)), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, width,
                           0xFFFFFFFFu / width + 1u, reinterpret_cast<uint4 *>(out));
    else if (!misaligned(out, 16))
        hipLaunchKernelGGL(onehot_kernel16<false>, dim3((unsigned)((n + 63) / 64)), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, width,
                           0xFFFFFFFFu / width + 1u, reinterpret_cast<uint4 *>(out));
    else
        hipLaunchKernelGGL(onehot_kernel, dim3((unsigned)((n + 63) / 64)), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, width,
                           reinterpret_cast<uint32_t *>(out));
    return launched("env2048_encode_onehot launch");
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
    hipLaunchKernelGGL(status_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, boards, done, legal, obs,
                       max_exp, n_empty, obs != nullptr);
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
