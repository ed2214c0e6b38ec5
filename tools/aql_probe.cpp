// aql_probe.cpp -- what does a dependent chain of step launches cost when the AQL packets are written by hand?
//
// The headline workload is K dependent launches of step_kernel over one batch.  Through HIP each launch costs 3.5-4 us of host
// time (so one thread cannot feed two half-batch chains) and carries whatever fences the runtime puts into the packet header.
// This probe loads the SAME code object (step_kernels.hip compiled --offload-device-only) through ROCr, creates its own user-mode
// queues and writes the kernel-dispatch packets itself:
//   * host cost per packet;
//   * period of a dependent chain (barrier bit set) with acquire / release fence scopes system / agent / none on the packets
//     between the first and the last (the first always acquires, the last always releases at system scope);
//   * two half-batch chains on two queues fed by ONE thread;
//   * argument blocks in device memory or in the kernarg pool of the host;
//   * every variant is checked bit for bit against the library's own env2048_step_sequence on the same boards.
// Build: tools/build_aql_probe.sh; run: tools/aql_probe <step_kernels.hsaco> [n] [K].  Measurement tool, not part of the product.

#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "env2048.h"
#include "tiling.hpp"

using env2048::block_boards;

#define HIPCHECK(x)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                         \
        }                                                                                    \
    } while (0)
#define HSACHECK(x)                                                              \
    do {                                                                         \
        hsa_status_t s_ = (x);                                                   \
        if (s_ != HSA_STATUS_SUCCESS) {                                          \
            const char *m_ = nullptr;                                            \
            hsa_status_string(s_, &m_);                                          \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m_ ? m_ : "?"); \
            exit(3);                                                             \
        }                                                                        \
    } while (0)
#define ENVCHECK(x)                                                                         \
    do {                                                                                    \
        int r_ = (x);                                                                       \
        if (r_ != 0) {                                                                      \
            fprintf(stderr, "%s:%d %s -> %d %s\n", __FILE__, __LINE__, #x, r_, env2048_last_error()); \
            exit(4);                                                                        \
        }                                                                                   \
    } while (0)

// the explicit arguments of step_kernel<EXTRAS, RANDOM, BPL> as the code object's metadata lays them out (104 bytes)
struct StepKernarg {
    uint64_t boards, actions, n, env_base;
    uint32_t flags, seed_lo, seed_hi, step_lo, step_hi, pad;
    uint64_t reward, done, legal, obs, max_exp, step_counter;
};
static_assert(sizeof(StepKernarg) == 104 && offsetof(StepKernarg, reward) == 56 && offsetof(StepKernarg, step_counter) == 96, "kernarg layout");
// ... and of env2048_step_resident_* (104 bytes; csrc/direct_dispatch.hpp is the product's user of these kernels)
struct ResidentKernarg {
    uint64_t boards, actions, n, env_base;
    uint32_t flags, seed_lo, seed_hi, step_lo, step_hi, pad;
    uint64_t reward, done;
    uint32_t mode, pad2;  // bit 0: first launch of a chain (records every workgroup's XCD, snapshots the boards)
    uint64_t owner, placement_error, snapshot;
};
static_assert(sizeof(ResidentKernarg) == 104 && offsetof(ResidentKernarg, mode) == 72 && offsetof(ResidentKernarg, snapshot) == 96, "kernarg layout");
constexpr size_t KERNARG_STRIDE = 128;

static hsa_agent_t g_gpu{};
static hsa_amd_memory_pool_t g_kernarg_pool{};
static bool g_have_gpu = false, g_have_pool = false;

static hsa_status_t find_gpu(hsa_agent_t agent, void *)
{
    hsa_device_type_t type;
    if (hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) == HSA_STATUS_SUCCESS && type == HSA_DEVICE_TYPE_GPU && !g_have_gpu) {
        g_gpu = agent;
        g_have_gpu = true;
    }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t find_kernarg_pool(hsa_amd_memory_pool_t pool, void *)
{
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_pool) {
        g_kernarg_pool = pool;
        g_have_pool = true;
    }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t find_cpu_pool(hsa_agent_t agent, void *)
{
    hsa_device_type_t type;
    if (hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) == HSA_STATUS_SUCCESS && type == HSA_DEVICE_TYPE_CPU && !g_have_pool)
        hsa_amd_agent_iterate_memory_pools(agent, find_kernarg_pool, nullptr);
    return HSA_STATUS_SUCCESS;
}

struct Kernel {
    uint64_t object = 0;
    uint32_t group = 0, priv = 0, kernarg = 0;
};

static Kernel get_kernel(hsa_executable_t exe, const char *name)
{
    hsa_executable_symbol_t sym;
    HSACHECK(hsa_executable_get_symbol_by_name(exe, name, &g_gpu, &sym));
    Kernel k;
    HSACHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    HSACHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    HSACHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    HSACHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    return k;
}

enum Fence { NONE = HSA_FENCE_SCOPE_NONE, AGENT = HSA_FENCE_SCOPE_AGENT, SYSTEM = HSA_FENCE_SCOPE_SYSTEM };

static inline uint16_t header(bool barrier, int acquire, int release)
{
    return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                      (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

// One kernel-dispatch packet into `q` (single producer): body first, then header + setup as one 32-bit release store, then the doorbell.
static inline void dispatch(hsa_queue_t *q, const Kernel &k, uint32_t blocks, uint64_t kernarg_address, uint16_t hdr, hsa_signal_t completion)
{
    const uint64_t index = hsa_queue_load_write_index_relaxed(q);
    while (index - hsa_queue_load_read_index_scacquire(q) >= q->size) {
    }
    hsa_kernel_dispatch_packet_t *p = reinterpret_cast<hsa_kernel_dispatch_packet_t *>(q->base_address) + (index & (q->size - 1));
    p->workgroup_size_x = (uint16_t)env2048::BLOCK;
    p->workgroup_size_y = 1;
    p->workgroup_size_z = 1;
    p->reserved0 = 0;
    p->grid_size_x = blocks * (uint32_t)env2048::BLOCK;
    p->grid_size_y = 1;
    p->grid_size_z = 1;
    p->private_segment_size = k.priv;
    p->group_segment_size = k.group;
    p->kernel_object = k.object;
    p->kernarg_address = reinterpret_cast<void *>(kernarg_address);
    p->reserved2 = 0;
    p->completion_signal = completion;
    const uint32_t word = (uint32_t)hdr | ((uint32_t)(1u << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16);
    __atomic_store_n(reinterpret_cast<uint32_t *>(p), word, __ATOMIC_RELEASE);
    hsa_queue_store_write_index_relaxed(q, index + 1);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)index);
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Buffers {
    uint64_t n;
    uint8_t *boards, *start, *ref_boards, *actions[8], *done;
    int32_t *reward;
};

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s step_kernels.hsaco [n = 1048576] [K = 200] [reps = 5]\n", argv[0]);
        return 1;
    }
    const uint64_t n = argc > 2 ? strtoull(argv[2], nullptr, 0) : (1ull << 20);
    const uint32_t K = argc > 3 ? (uint32_t)atoi(argv[3]) : 200u;
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const bool small = env2048::plan_small_tiling(n, 0);
    const int bpl = small ? 1 : env2048::BOARDS_PER_LANE;
    const uint64_t seed = 0x2048, step0 = 1000;
    if (n % block_boards(bpl) != 0) {
        fprintf(stderr, "n must be a multiple of %d boards (whole blocks)\n", block_boards(bpl));
        return 1;
    }

    HIPCHECK(hipSetDevice(0));
    Buffers b{};
    b.n = n;
    HIPCHECK(hipMalloc(&b.boards, 16 * n));
    HIPCHECK(hipMalloc(&b.start, 16 * n));
    HIPCHECK(hipMalloc(&b.ref_boards, 16 * n));
    HIPCHECK(hipMalloc(&b.reward, 4 * n));
    HIPCHECK(hipMalloc(&b.done, n));
    std::vector<uint8_t> host_actions(n);
    srand(1);
    for (int r = 0; r < 8; ++r) {
        HIPCHECK(hipMalloc(&b.actions[r], n));
        for (uint64_t i = 0; i < n; ++i) host_actions[i] = (uint8_t)(rand() & 3);
        HIPCHECK(hipMemcpy(b.actions[r], host_actions.data(), n, hipMemcpyHostToDevice));
    }
    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));
    // mid-game boards: reset + 64 steps with the moves drawn in the kernel
    ENVCHECK(env2048_reset(0, stream, n, 0, b.boards, b.done, seed, 0));
    for (int t = 0; t < 64; ++t)
        ENVCHECK(env2048_step(0, stream, n, 0, b.boards, nullptr, b.reward, b.done, nullptr, nullptr, seed, 1 + t,
                              ENV2048_FLAG_AUTO_RESET | ENV2048_FLAG_RANDOM_ACTIONS));
    HIPCHECK(hipStreamSynchronize(stream));
    HIPCHECK(hipMemcpy(b.start, b.boards, 16 * n, hipMemcpyDeviceToDevice));

    // reference: the library's own sequence (one chain of HIP launches), timed the way bench.py times it
    std::vector<const uint8_t *> table(K);
    for (uint32_t t = 0; t < K; ++t) table[t] = b.actions[t & 7];
    const uint32_t flags = ENV2048_FLAG_AUTO_RESET;
    std::vector<uint8_t> ref(16 * n), got(16 * n);
    std::vector<int32_t> ref_reward(n), got_reward(n);
    std::vector<uint8_t> ref_done(n), got_done(n);
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    for (int chains = 1; chains <= 2; ++chains) {
        double best_wall = 1e30, best_event = 1e30;
        for (int r = 0; r < reps; ++r) {
            HIPCHECK(hipMemcpyAsync(b.boards, b.start, 16 * n, hipMemcpyDeviceToDevice, stream));
            HIPCHECK(hipStreamSynchronize(stream));
            const double t0 = now_us();
            HIPCHECK(hipEventRecord(e0, stream));
            ENVCHECK(env2048_step_sequence(0, stream, n, 0, b.boards, table.data(), b.reward, b.done, nullptr, nullptr, 0, seed, step0, K,
                                           flags | (chains == 1 ? ENV2048_FLAG_SINGLE_CHAIN : 0u)));
            HIPCHECK(hipEventRecord(e1, stream));
            HIPCHECK(hipStreamSynchronize(stream));
            const double t1 = now_us();
            float ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
            best_wall = std::min(best_wall, (t1 - t0) / K);
            best_event = std::min(best_event, ms * 1e3 / K);
        }
        printf("hip   chains=%d                          K=%u  wall %.3f us/step  event %.3f us/step\n", chains, K, best_wall, best_event);
    }
    HIPCHECK(hipMemcpy(ref.data(), b.boards, 16 * n, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(ref_reward.data(), b.reward, 4 * n, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(ref_done.data(), b.done, n, hipMemcpyDeviceToHost));

    // ---- ROCr: agent, queues, code object --------------------------------------------------------------------------------
    HSACHECK(hsa_init());
    HSACHECK(hsa_iterate_agents(find_gpu, nullptr));
    HSACHECK(hsa_iterate_agents(find_cpu_pool, nullptr));
    if (!g_have_gpu || !g_have_pool) {
        fprintf(stderr, "no GPU agent / kernarg pool\n");
        return 5;
    }
    char agent_name[64] = {};
    hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, agent_name);
    hsa_queue_t *queue[2];
    for (int i = 0; i < 2; ++i) HSACHECK(hsa_queue_create(g_gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &queue[i]));
    FILE *f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 6;
    }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> image(size);
    if (fread(image.data(), 1, size, f) != (size_t)size) return 6;
    fclose(f);
    hsa_code_object_reader_t reader;
    hsa_executable_t exe;
    HSACHECK(hsa_code_object_reader_create_from_memory(image.data(), size, &reader));
    hsa_profile_t profile = HSA_PROFILE_BASE;
    hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_PROFILE, &profile);
    HSACHECK(hsa_executable_create_alt(profile, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HSACHECK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HSACHECK(hsa_executable_freeze(exe, nullptr));
    const std::string name = std::string("_ZN12_GLOBAL__N_111step_kernelILb0ELb0ELi") + (small ? "1" : "4") + "EEEvPhPKhmmjjjjjPiS1_S1_PvS1_Py.kd";
    const bool resident_kernel = getenv("AQL_RESIDENT_KERNEL") && atoi(getenv("AQL_RESIDENT_KERNEL"));  // env2048_step_resident_*: sc1 board loads
    const Kernel kernel = get_kernel(exe, resident_kernel ? (small ? "env2048_step_resident_small.kd" : "env2048_step_resident_large.kd") : name.c_str());
    printf("agent %s  kernel object %#llx  group %u  private %u  kernarg %u  (n = %llu, %d boards per lane, K = %u)\n", agent_name,
           (unsigned long long)kernel.object, kernel.group, kernel.priv, kernel.kernarg, (unsigned long long)n, bpl, K);
    if (kernel.kernarg != (resident_kernel ? sizeof(ResidentKernarg) : sizeof(StepKernarg)) || kernel.priv != 0) {
        fprintf(stderr, "unexpected kernel resources\n");
        return 7;
    }

    // argument blocks: [cut][part][K] for the batch cut into 1 ... MAX_PARTS contiguous parts of whole blocks, in device memory and
    // in the host pool.  AQL_ROWS=1: reward / done of step t go to row t (K x n outputs, as a recording loop has them) instead of
    // being overwritten in place.
    const bool write_back = getenv("AQL_WRITE_BACK") && atoi(getenv("AQL_WRITE_BACK"));  // A/B: plain stores (lines stay in the XCD's L2)
    const bool rows = getenv("AQL_ROWS") && atoi(getenv("AQL_ROWS"));
    uint32_t kflags = flags | env2048::INTERNAL_FLAG_WIDE_OUTPUTS;
    if (env2048::plan_write_through(n, 0) && !write_back) kflags |= env2048::INTERNAL_FLAG_WRITE_THROUGH;
    printf("stores: %s; reward / done: %s\n", (kflags & env2048::INTERNAL_FLAG_WRITE_THROUGH) ? "write-through (sc1)" : "write-back",
           rows ? "one row per step" : "overwritten in place");
    int32_t *reward_rows = b.reward;
    uint8_t *done_rows = b.done;
    if (rows) {
        HIPCHECK(hipMalloc(&reward_rows, 4 * n * (size_t)K));
        HIPCHECK(hipMalloc(&done_rows, n * (size_t)K));
    }
    constexpr int MAX_PARTS = 4;
    hsa_queue_t *queues[MAX_PARTS] = {queue[0], queue[1], nullptr, nullptr};
    for (int i = 2; i < MAX_PARTS; ++i) HSACHECK(hsa_queue_create(g_gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &queues[i]));
    // AQL_PAIR="a,b": the 2-chain variants run on queues a and b of EIGHT queues created in order (do two queues created four apart
    // share a hardware pipe?  profiles/r05_j_*)
    if (const char *pair = getenv("AQL_PAIR")) {
        hsa_queue_t *all[8] = {queues[0], queues[1], queues[2], queues[3], nullptr, nullptr, nullptr, nullptr};
        for (int i = 4; i < 8; ++i) HSACHECK(hsa_queue_create(g_gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &all[i]));
        int a = 0, b = 1;
        sscanf(pair, "%d,%d", &a, &b);
        queues[0] = all[a & 7];
        queues[1] = all[b & 7];
        printf("2-chain variants on queues %d and %d of 8\n", a & 7, b & 7);
    }
    uint32_t *placement_error = nullptr;  // pinned host word the resident kernels report a misplaced workgroup into
    HSACHECK(hsa_amd_memory_pool_allocate(g_kernarg_pool, 4096, 0, (void **)&placement_error));
    HSACHECK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, placement_error));
    *placement_error = 0;
    uint32_t *owner_table = nullptr;  // [part][workgroup]: the XCD the first launch of a chain found each workgroup on
    uint8_t *snapshot = nullptr;
    HIPCHECK(hipMalloc(&owner_table, MAX_PARTS * 4096 * 4));
    HIPCHECK(hipMalloc(&snapshot, 16 * n));
    const uint64_t total_blocks = n / block_boards(bpl);
    auto part_first_block = [&](int parts, int p) { return total_blocks * (uint64_t)p / (uint64_t)parts; };
    auto table_slot = [&](int parts, int p, uint32_t t) { return ((size_t)(parts * (parts - 1) / 2 + p) * K + t) * KERNARG_STRIDE; };
    std::vector<uint8_t> blocks_host(table_slot(MAX_PARTS + 1, 0, 0), 0);
    for (int parts = 1; parts <= MAX_PARTS; ++parts)
        for (int p = 0; p < parts; ++p)
            for (uint32_t t = 0; t < K; ++t) {
                const uint64_t off = part_first_block(parts, p) * block_boards(bpl);
                const uint64_t count = part_first_block(parts, p + 1) * block_boards(bpl) - off;
                const uint64_t row = rows ? (uint64_t)t * n : 0;
                StepKernarg a{};
                a.boards = (uint64_t)(b.boards + 16 * off);
                a.actions = (uint64_t)(b.actions[t & 7] + off);
                a.n = count;
                a.env_base = off;
                a.flags = kflags;
                a.seed_lo = (uint32_t)seed;
                a.seed_hi = (uint32_t)(seed >> 32);
                a.step_lo = (uint32_t)(step0 + t);
                a.step_hi = (uint32_t)((step0 + t) >> 32);
                a.reward = (uint64_t)(reward_rows + row + off);
                a.done = (uint64_t)(done_rows + row + off);
                if (resident_kernel) {
                    ResidentKernarg ra{};
                    memcpy(&ra, &a, offsetof(ResidentKernarg, mode));  // (the two layouts agree up to `done`)
                    ra.mode = t == 0 ? 1u : 0u;
                    ra.owner = (uint64_t)(owner_table + 4096 * p);
                    ra.placement_error = (uint64_t)placement_error;
                    ra.snapshot = (uint64_t)(snapshot + 16 * off);
                    memcpy(blocks_host.data() + table_slot(parts, p, t), &ra, sizeof ra);
                } else
                    memcpy(blocks_host.data() + table_slot(parts, p, t), &a, sizeof a);
            }
    uint8_t *blocks_dev = nullptr;
    void *blocks_pool = nullptr;
    HIPCHECK(hipMalloc(&blocks_dev, blocks_host.size()));
    HIPCHECK(hipMemcpy(blocks_dev, blocks_host.data(), blocks_host.size(), hipMemcpyHostToDevice));
    HSACHECK(hsa_amd_memory_pool_allocate(g_kernarg_pool, blocks_host.size(), 0, &blocks_pool));
    HSACHECK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, blocks_pool));
    memcpy(blocks_pool, blocks_host.data(), blocks_host.size());

    hsa_signal_t done_signal[MAX_PARTS];
    for (int i = 0; i < MAX_PARTS; ++i) HSACHECK(hsa_signal_create(1, 0, nullptr, &done_signal[i]));

    auto restore = [&]() {
        HIPCHECK(hipMemcpyAsync(b.boards, b.start, 16 * n, hipMemcpyDeviceToDevice, stream));
        HIPCHECK(hipMemsetAsync(reward_rows, 0xff, 4 * n * (rows ? (size_t)K : 1), stream));
        HIPCHECK(hipMemsetAsync(done_rows, 0xff, n * (rows ? (size_t)K : 1), stream));
        HIPCHECK(hipStreamSynchronize(stream));
    };
    auto check = [&](const char *what) {  // (the LAST row of reward / done is what the reference run left behind)
        const uint64_t row = rows ? (uint64_t)(K - 1) * n : 0;
        HIPCHECK(hipMemcpy(got.data(), b.boards, 16 * n, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(got_reward.data(), reward_rows + row, 4 * n, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(got_done.data(), done_rows + row, n, hipMemcpyDeviceToHost));
        uint64_t bad = 0;
        for (uint64_t i = 0; i < n; ++i)
            bad += memcmp(&got[16 * i], &ref[16 * i], 16) != 0 || got_reward[i] != ref_reward[i] || got_done[i] != ref_done[i];
        if (bad) printf("    !! %s: %llu of %llu environments differ from the library's sequence\n", what, (unsigned long long)bad, (unsigned long long)n);
        return bad;
    };

    // One timed variant: the batch as `parts` contiguous parts, part p a dependent chain (barrier bit) on queues[p]; the packets are
    // written by this thread, step by step, part by part.  The first packet of every chain acquires and the last releases at
    // system scope; `mid` is the fence scope of everything between.
    auto run_variant = [&](int parts, int mid, bool dev_args) {
        char label[96];
        snprintf(label, sizeof label, "%d chain%s fences %-6s args %s", parts, parts > 1 ? "s" : " ", mid == SYSTEM ? "system" : mid == AGENT ? "agent" : "none",
                 dev_args ? "device" : "host");
        if (total_blocks < (uint64_t)parts) return;
        double best_wall = 1e30, best_enqueue = 1e30;
        std::vector<double> walls;
        uint64_t bad = 0;
        for (int r = 0; r < reps; ++r) {
            restore();
            for (int i = 0; i < parts; ++i) hsa_signal_store_relaxed(done_signal[i], 1);
            const uint64_t base = (uint64_t)(dev_args ? (void *)blocks_dev : blocks_pool);
            const hsa_signal_t null_signal{0};
            const double t0 = now_us();
            for (uint32_t t = 0; t < K; ++t) {
                const bool first = t == 0, last = t + 1 == K;
                const uint16_t hdr = header(true, first ? SYSTEM : mid, last ? SYSTEM : mid);
                for (int p = 0; p < parts; ++p)
                    dispatch(queues[p], kernel, (uint32_t)(part_first_block(parts, p + 1) - part_first_block(parts, p)), base + table_slot(parts, p, t), hdr,
                             last ? done_signal[p] : null_signal);
            }
            const double t1 = now_us();
            for (int i = 0; i < parts; ++i)
                while (hsa_signal_wait_scacquire(done_signal[i], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {
                }
            const double t2 = now_us();
            walls.push_back((t2 - t0) / K);
            best_wall = std::min(best_wall, (t2 - t0) / K);
            best_enqueue = std::min(best_enqueue, (t1 - t0) / ((double)K * parts));
            if (r == 0) bad = check(label);
        }
        std::sort(walls.begin(), walls.end());
        printf("aql   %-42s K=%u  wall best %.3f median %.3f us/step  host %.3f us/packet  %s%s\n", label, K, best_wall, walls[walls.size() / 2], best_enqueue,
               bad ? "MISMATCH" : "bit-exact", *placement_error ? "  PLACEMENT ERROR" : "");
        fflush(stdout);
    };

    if (!getenv("AQL_SKIP_SLOW")) {
        run_variant(1, SYSTEM, true);
        run_variant(1, AGENT, false);
    }
    for (int parts = 1; parts <= (getenv("AQL_PAIR") ? 2 : MAX_PARTS); ++parts) {
        run_variant(parts, AGENT, true);
        run_variant(parts, NONE, true);
    }
    if (getenv("AQL_PAIR")) return 0;

    // ---- which XCD does workgroup k of a dispatch run on, queue by queue? --------------------------------------------------
    // (CU masks do not pin a queue to an XCD -- hsa_amd_queue_cu_set_mask is applied inside every XCD alike, profiles/r05_b_*; the
    // placement is a property of the QUEUE: workgroup k -> XCD (k + s) % 8 with s fixed when the queue is created.)
    const std::string aux_path = argc > 5 ? argv[5] : std::string(argv[1]).substr(0, std::string(argv[1]).find_last_of('/') + 1) + "aql_aux.hsaco";
    f = fopen(aux_path.c_str(), "rb");
    if (!f) {
        perror(aux_path.c_str());
        return 6;
    }
    fseek(f, 0, SEEK_END);
    const long aux_size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> aux_image(aux_size);
    if (fread(aux_image.data(), 1, aux_size, f) != (size_t)aux_size) return 6;
    fclose(f);
    hsa_code_object_reader_t aux_reader;
    hsa_executable_t aux_exe;
    HSACHECK(hsa_code_object_reader_create_from_memory(aux_image.data(), aux_size, &aux_reader));
    HSACHECK(hsa_executable_create_alt(profile, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &aux_exe));
    HSACHECK(hsa_executable_load_agent_code_object(aux_exe, g_gpu, aux_reader, nullptr, nullptr));
    HSACHECK(hsa_executable_freeze(aux_exe, nullptr));
    const Kernel xcc_kernel = get_kernel(aux_exe, "xcc_probe_kernel.kd");
    constexpr uint32_t PROBE_BLOCKS = 512;
    uint32_t *xcc_out = nullptr;
    uint8_t *xcc_args = nullptr;
    HIPCHECK(hipMalloc(&xcc_out, PROBE_BLOCKS * 4));
    HIPCHECK(hipMalloc(&xcc_args, 1024));
    {
        std::vector<uint8_t> args(1024, 0);
        memcpy(args.data(), &xcc_out, 8);
        HIPCHECK(hipMemcpy(xcc_args, args.data(), 1024, hipMemcpyHostToDevice));
    }
    auto xcc_of_blocks = [&](hsa_queue_t *q, uint32_t blocks) {
        HIPCHECK(hipMemset(xcc_out, 0xff, PROBE_BLOCKS * 4));
        HIPCHECK(hipDeviceSynchronize());
        hsa_signal_store_relaxed(done_signal[0], 1);
        dispatch(q, xcc_kernel, blocks, (uint64_t)xcc_args, header(true, SYSTEM, SYSTEM), done_signal[0]);
        while (hsa_signal_wait_scacquire(done_signal[0], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {
        }
        std::vector<uint32_t> out(blocks);
        HIPCHECK(hipMemcpy(out.data(), xcc_out, blocks * 4, hipMemcpyDeviceToHost));
        for (uint32_t &v : out) v &= 0xfu;
        return out;
    };
    for (int qi = 0; qi < MAX_PARTS; ++qi)
        for (uint32_t blocks : {512u, 13u, 100u, 512u}) {
            const std::vector<uint32_t> x = xcc_of_blocks(queues[qi], blocks);
            bool round_robin = true;
            for (uint32_t k = 0; k < blocks; ++k) round_robin &= x[k] == (x[0] + k) % 8;
            printf("queue %d, %3u workgroups: XCC_ID of workgroups 0..11 =", qi, blocks);
            for (uint32_t k = 0; k < 12 && k < blocks; ++k) printf(" %u", x[k]);
            printf("  -> workgroup k on XCD (k + %u) %% 8 for every k: %s\n", x[0], round_robin ? "yes" : "NO");
        }
    // ---- argument blocks in device memory the HOST writes directly (large BAR): no copy, one read-back as the flush --------
    {
        hsa_amd_memory_pool_t local{};
        bool have_local = false;
        struct Ctx {
            hsa_amd_memory_pool_t *pool;
            bool *have;
        } ctx{&local, &have_local};
        hsa_amd_agent_iterate_memory_pools(
            g_gpu,
            [](hsa_amd_memory_pool_t pool, void *data) {
                Ctx *c = static_cast<Ctx *>(data);
                hsa_amd_segment_t seg;
                uint32_t flags = 0;
                bool alloc = false;
                hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
                hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
                hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
                if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !*c->have) {
                    *c->pool = pool;
                    *c->have = true;
                }
                return HSA_STATUS_SUCCESS;
            },
            &ctx);
        hsa_agent_t cpu{};
        bool have_cpu = false;
        struct CpuCtx {
            hsa_agent_t *agent;
            bool *have;
        } cctx{&cpu, &have_cpu};
        hsa_iterate_agents(
            [](hsa_agent_t agent, void *data) {
                CpuCtx *c = static_cast<CpuCtx *>(data);
                hsa_device_type_t type;
                if (hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) == HSA_STATUS_SUCCESS && type == HSA_DEVICE_TYPE_CPU && !*c->have) {
                    *c->agent = agent;
                    *c->have = true;
                }
                return HSA_STATUS_SUCCESS;
            },
            &cctx);
        void *bar = nullptr;
        hsa_amd_memory_pool_access_t access = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
        if (have_local && have_cpu) hsa_amd_agent_memory_pool_get_info(cpu, local, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &access);
        printf("device-local pool: %s, host access: %s\n", have_local ? "found" : "none",
               access == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED ? "never" : access == HSA_AMD_MEMORY_POOL_ACCESS_ALLOWED_BY_DEFAULT ? "by default" : "on request");
        if (have_local && have_cpu && access != HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED &&
            hsa_amd_memory_pool_allocate(local, blocks_host.size(), 0, &bar) == HSA_STATUS_SUCCESS) {
            hsa_agent_t both[2] = {cpu, g_gpu};
            const hsa_status_t st = hsa_amd_agents_allow_access(2, both, nullptr, bar);
            if (st == HSA_STATUS_SUCCESS) {
                uint8_t *saved = blocks_dev;
                double fill = 1e30;
                for (int r = 0; r < 5; ++r) {  // the host fills a 2-chain, K-step table and reads one word back
                    const double t0 = now_us();
                    memcpy((uint8_t *)bar + table_slot(2, 0, 0), blocks_host.data() + table_slot(2, 0, 0), 2 * (size_t)K * KERNARG_STRIDE);
                    __atomic_thread_fence(__ATOMIC_SEQ_CST);
                    volatile uint64_t back = *(volatile uint64_t *)((uint8_t *)bar + table_slot(2, 1, K - 1));
                    (void)back;
                    fill = std::min(fill, now_us() - t0);
                }
                memcpy(bar, blocks_host.data(), blocks_host.size());
                __atomic_thread_fence(__ATOMIC_SEQ_CST);
                volatile uint64_t back = *(volatile uint64_t *)((uint8_t *)bar + blocks_host.size() - KERNARG_STRIDE);
                (void)back;
                printf("host-written device memory: %zu bytes of argument blocks + read-back in %.2f us\n", 2 * (size_t)K * KERNARG_STRIDE, fill);
                blocks_dev = (uint8_t *)bar;
                printf("(argument blocks written by the host into device memory:)\n");
                run_variant(2, NONE, true);
                run_variant(2, AGENT, true);
                blocks_dev = saved;
            } else {
                printf("hsa_amd_agents_allow_access(cpu + gpu) on device-local memory failed\n");
            }
        }
    }

    // ---- interop with a HIP stream ---------------------------------------------------------------------------------------------
    // join: the chain's last packet stores a serial into signal memory, the HIP stream waits for it (hipStreamWaitValue64)
    // fork: the HIP stream writes 0 into an HSA signal (hipStreamWriteValue64 on its value pointer), the chain starts behind a
    //       barrier-AND packet on that signal
    {
        const Kernel store_kernel = get_kernel(aux_exe, "store_u64_kernel.kd");
        int can_wait = 0;
        (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
        uint64_t *words = nullptr;
        const hipError_t we = hipExtMallocWithFlags((void **)&words, 4096, hipMallocSignalMemory);
        printf("hipDeviceAttributeCanUseStreamWaitValue = %d, hipExtMallocWithFlags(hipMallocSignalMemory) -> %s\n", can_wait, hipGetErrorString(we));
        hsa_signal_t fork_signal;
        HSACHECK(hsa_signal_create(1, 0, nullptr, &fork_signal));
        volatile hsa_signal_value_t *fork_value = nullptr;
        const hsa_status_t vp = hsa_amd_signal_value_pointer(fork_signal, &fork_value);
        printf("hsa_amd_signal_value_pointer -> %s (%p)\n", vp == HSA_STATUS_SUCCESS ? "ok" : "failed", (void *)fork_value);
        if (we == hipSuccess && can_wait) {
            HIPCHECK(hipMemset(words, 0, 4096));
            uint8_t *store_args = nullptr;
            HIPCHECK(hipMalloc(&store_args, 4096));
            const int parts = 2;
            const uint32_t lead = 3;  // steps the HIP stream runs itself before it hands over
            auto barrier_and = [&](hsa_queue_t *q, hsa_signal_t dep) {
                const uint64_t index = hsa_queue_load_write_index_relaxed(q);
                hsa_barrier_and_packet_t *p = reinterpret_cast<hsa_barrier_and_packet_t *>(q->base_address) + (index & (q->size - 1));
                memset(reinterpret_cast<uint8_t *>(p) + 4, 0, sizeof *p - 4);
                p->dep_signal[0] = dep;
                const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER));
                __atomic_store_n(reinterpret_cast<uint32_t *>(p), (uint32_t)hdr, __ATOMIC_RELEASE);
                hsa_queue_store_write_index_relaxed(q, index + 1);
                hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)index);
            };
            for (int mode = 0; mode < 2; ++mode) {  // 0: join only (the stream is idle at the fork); 1: fork + join
                double best = 1e30, best_return = 1e30;
                uint64_t bad = 0;
                bool fork_ok = true;
                for (int r = 0; r < reps; ++r) {
                    restore();
                    const uint64_t serial = 100 + (uint64_t)(mode * reps + r);
                    std::vector<uint8_t> args(4096, 0);
                    for (int p = 0; p < parts; ++p) {
                        uint64_t *w = words + 16 * p;
                        memcpy(args.data() + 256 * p, &w, 8);
                        memcpy(args.data() + 256 * p + 8, &serial, 8);
                    }
                    HIPCHECK(hipMemcpy(store_args, args.data(), 4096, hipMemcpyHostToDevice));
                    hsa_signal_store_relaxed(fork_signal, 1);
                    const hsa_signal_t null_signal{0};
                    const double t0 = now_us();
                    uint32_t first_step = 0;
                    if (mode == 1) {
                        first_step = lead;
                        for (uint32_t t = 0; t < lead; ++t)
                            ENVCHECK(env2048_step(0, stream, n, 0, b.boards, b.actions[t & 7], reward_rows + (rows ? (uint64_t)t * n : 0),
                                                  done_rows + (rows ? (uint64_t)t * n : 0), nullptr, nullptr, seed, step0 + t, flags));
                        const hipError_t fe = hipStreamWriteValue64(stream, (void *)fork_value, 0, 0);
                        if (fe != hipSuccess) {
                            if (fork_ok) printf("hipStreamWriteValue64 on the HSA signal's value pointer -> %s\n", hipGetErrorString(fe));
                            fork_ok = false;
                            (void)hipGetLastError();
                            hsa_signal_store_screlease(fork_signal, 0);
                            HIPCHECK(hipStreamSynchronize(stream));
                        }
                        for (int p = 0; p < parts; ++p) barrier_and(queues[p], fork_signal);
                    }
                    for (uint32_t t = first_step; t < K; ++t) {
                        const bool first = t == first_step, last = t + 1 == K;
                        const uint16_t hdr = header(true, first ? SYSTEM : NONE, last ? SYSTEM : NONE);
                        for (int p = 0; p < parts; ++p)
                            dispatch(queues[p], kernel, (uint32_t)(part_first_block(parts, p + 1) - part_first_block(parts, p)),
                                     (uint64_t)blocks_dev + table_slot(parts, p, t), hdr, null_signal);
                    }
                    for (int p = 0; p < parts; ++p) {  // (one 64-thread workgroup: grid = 1 block of BLOCK threads is fine, thread 0 stores)
                        dispatch(queues[p], store_kernel, 1, (uint64_t)store_args + 256 * p, header(true, NONE, SYSTEM), null_signal);
                        HIPCHECK(hipStreamWaitValue64(stream, words + 16 * p, serial, hipStreamWaitValueGte, ~0ull));
                    }
                    const double t1 = now_us();
                    HIPCHECK(hipStreamSynchronize(stream));
                    const double t2 = now_us();
                    best = std::min(best, (t2 - t0) / K);
                    best_return = std::min(best_return, (t1 - t0) / K);
                    if (r == 0) bad = check(mode ? "fork + join" : "join");
                }
                printf("aql   2 chains resident, %s with the HIP stream: K=%u  wall %.3f us/step (call returns after %.3f us/step)  %s%s\n",
                       mode ? "fork (3 HIP steps first) + join" : "join", K, best, best_return, bad ? "MISMATCH" : "bit-exact",
                       mode && !fork_ok ? "  [fork by host wait]" : "");
                fflush(stdout);
            }
        }
        hsa_signal_destroy(fork_signal);
    }
    for (int i = 0; i < MAX_PARTS; ++i) {
        hsa_signal_destroy(done_signal[i]);
        hsa_queue_destroy(queues[i]);
    }
    hsa_executable_destroy(exe);
    hsa_code_object_reader_destroy(reader);
    hsa_shut_down();
    return 0;
}
