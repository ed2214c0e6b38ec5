// Experimental variants (included by microbench.hip): 4 boards per lane, shared Philox block,
// per-lane v_perm selectors from an LDS table.
#pragma once
#include "board_ops.hpp"
using namespace env2048;

__device__ __constant__ uint32_t DIR_TABLE_C[32] = {
    0x06020400u, 0x07030501u, 0x05040100u, 0x07060302u, 0x06020400u, 0x07030501u, 0x05040100u, 0x07060302u,  // left
    0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u,  // up
    0x05010703u, 0x04000602u, 0x05040100u, 0x07060302u, 0x02060004u, 0x03070105u, 0x01000504u, 0x03020706u,  // right
    0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u,  // down
};

__device__ __forceinline__ void afterstate_sel4(Board &bd, const uint4 &f, const uint4 &inv, uint32_t &score, bool &changed)
{
    uint32_t t0 = byte_perm(bd.r1, bd.r0, f.x), t2 = byte_perm(bd.r3, bd.r2, f.x);
    uint32_t t1 = byte_perm(bd.r1, bd.r0, f.y), t3 = byte_perm(bd.r3, bd.r2, f.y);
    uint32_t p0 = byte_perm(t2, t0, f.z), p2 = byte_perm(t2, t0, f.w);
    uint32_t p1 = byte_perm(t3, t1, f.z), p3 = byte_perm(t3, t1, f.w);
    uint32_t q0 = p0, q1 = p1, q2 = p2, q3 = p3, changed_bits;
    score = slide_merge_planes(q0, q1, q2, q3, changed_bits);
    changed = changed_bits != 0u;
    t0 = byte_perm(q1, q0, inv.x); t2 = byte_perm(q3, q2, inv.x);
    t1 = byte_perm(q1, q0, inv.y); t3 = byte_perm(q3, q2, inv.y);
    bd.r0 = byte_perm(t2, t0, inv.z); bd.r2 = byte_perm(t2, t0, inv.w);
    bd.r1 = byte_perm(t3, t1, inv.z); bd.r3 = byte_perm(t3, t1, inv.w);
}

// one-word spawn: k = hi32(m*n), tile 4 iff lo32(m*n) < T, with m = w & ~3
__device__ __forceinline__ uint32_t spawn_tile1(Board &bd, uint32_t w, bool enable)
{
    uint32_t empt = empty_mask(bd);
    uint32_t n_empty = __popc(empt);
    uint64_t prod = (uint64_t)(w & ~3u) * n_empty;
    uint32_t bit = kth_empty(empt, (uint32_t)(prod >> 32));
    bit = enable ? bit : 0u;
    place_tile(bd, bit, (uint32_t)prod < SPAWN4_THRESHOLD);
    return n_empty - (enable ? 1u : 0u);
}

struct Args4 {
    uint64_t n;
    uint4 *boards;
    const uint8_t *actions;
    int32_t *reward;
    uint8_t *done;
    RngKey key;
    uint32_t flags;
};

// MODE bit0: shared Philox (1 block per 4 boards, 1 word per spawn); bit1: LDS selector table
template <int MODE, bool TILE>
__global__ __launch_bounds__(256) void k_four(Args4 a)
{
    __shared__ uint4 dir_lds[8];
    if (MODE & 2) {
        if (threadIdx.x < 8) dir_lds[threadIdx.x] = reinterpret_cast<const uint4 *>(DIR_TABLE_C)[threadIdx.x];
        __syncthreads();
    }
    const uint64_t lane_id = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // one lane = 4 boards
    uint64_t idx[4];
    if (TILE) {
        const uint64_t wave_base = (lane_id & ~63ull) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) idx[j] = wave_base + 64 * j + (lane_id & 63);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) idx[j] = lane_id * 4 + j;
    }
    if (idx[0] >= a.n) return;  // n is a multiple of 256 in the microbench
    Board bd[4];
    uint32_t act[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bd[j] = Board{a.boards[idx[j]].x, a.boards[idx[j]].y, a.boards[idx[j]].z, a.boards[idx[j]].w};
    if (TILE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) act[j] = a.actions[idx[j]];
    } else {
        uint32_t packed = reinterpret_cast<const uint32_t *>(a.actions)[lane_id];
#pragma unroll
        for (int j = 0; j < 4; ++j) act[j] = (packed >> (8 * j)) & 3u;
    }
    uint32_t w[4];
    if (MODE & 1) env_block(a.key, lane_id, 0u, w[0], w[1], w[2], w[3]);
    uint32_t rew[4];
    uint32_t dn = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t reward;
        bool changed, done;
        uint32_t empties;
        if (MODE & 2) {
            uint4 f = dir_lds[2 * (act[j] & 3)], inv = dir_lds[2 * (act[j] & 3) + 1];
            afterstate_sel4(bd[j], f, inv, reward, changed);
        } else {
            afterstate(bd[j], act[j] & 3u, reward, changed);
        }
        if (MODE & 1) {
            empties = spawn_tile1(bd[j], w[j], changed);
        } else {
            uint32_t x0, x1, x2, x3;
            env_block(a.key, idx[j], 0u, x0, x1, x2, x3);
            empties = spawn_tile(bd[j], x0, x1, changed);
        }
        done = (empties == 0u) && no_equal_neighbours(bd[j]);
        if ((a.flags & 1u) && done) {
            uint32_t x0, x1, x2, x3;
            env_block(a.key, idx[j], 1u, x0, x1, x2, x3);
            bd[j] = fresh_board(x0, x1, x2, x3);
        }
        rew[j] = reward;
        dn |= (done ? 1u : 0u) << (8 * j);
        a.boards[idx[j]] = make_uint4(bd[j].r0, bd[j].r1, bd[j].r2, bd[j].r3);
    }
    if (TILE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a.reward[idx[j]] = (int32_t)rew[j];
            a.done[idx[j]] = (uint8_t)(dn >> (8 * j));
        }
    } else {
        reinterpret_cast<uint4 *>(a.reward)[lane_id] = make_uint4(rew[0], rew[1], rew[2], rew[3]);
        reinterpret_cast<uint32_t *>(a.done)[lane_id] = dn;
    }
}

// 1 board per lane with the LDS selector table and/or a "quarter Philox" (cost probe: each lane
// still computes a full block but 4 lanes would share it; here we emulate the ALU saving by
// computing the block only on... not possible in SIMD -> only the LDS-dir variant is meaningful)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_v1_ldsdir(Args4 a)
{
    __shared__ uint4 dir_lds[8];
    if (threadIdx.x < 8) dir_lds[threadIdx.x] = reinterpret_cast<const uint4 *>(DIR_TABLE_C)[threadIdx.x];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    uint4 v = a.boards[i];
    Board bd{v.x, v.y, v.z, v.w};
    uint32_t w0, w1, w2, w3;
    env_block(a.key, i, 0u, w0, w1, w2, w3);
    uint32_t act = a.actions[i] & 3u;
    uint4 f = dir_lds[2 * act], inv = dir_lds[2 * act + 1];
    uint32_t reward;
    bool changed;
    afterstate_sel4(bd, f, inv, reward, changed);
    uint32_t empties = spawn_tile(bd, w0, w1, changed);
    bool done = (empties == 0u) && no_equal_neighbours(bd);
    if ((a.flags & 1u) && done) {
        env_block(a.key, i, 1u, w0, w1, w2, w3);
        bd = fresh_board(w0, w1, w2, w3);
    }
    a.boards[i] = make_uint4(bd.r0, bd.r1, bd.r2, bd.r3);
    a.reward[i] = (int32_t)reward;
    a.done[i] = done ? 1 : 0;
}
