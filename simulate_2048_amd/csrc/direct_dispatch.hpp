// direct_dispatch.hpp -- env2048_step_sequence as hand-written AQL packets on queues of the library's own (host code; HIP + ROCr).
//
// K dependent steps of one batch are K dependent launches.  Through a HIP stream every launch costs 3.5-4 us of host time (one
// thread cannot feed two half-batch chains: hence helper_thread.hpp) and carries the runtime's cache fences: on a chip whose
// eight XCD L2s are not coherent with each other, an agent-scope release writes an XCD's dirty lines back at the end of every
// kernel and an agent-scope acquire drops them at the start of the next, so the 16 MiB of boards of a 2^20-board batch cross
// the fabric twice per step although every tile is read and written by the same workgroup index launch after launch.
//
// Here the library loads the step kernels a second time as a bare code object through ROCr (the image is linked into this
// shared object, build.py), owns two user-mode queues per device and writes the kernel-dispatch packets itself:
//   * 0.13-0.25 us of host time per packet: the calling thread feeds both chains, no helper thread, no fork / join events;
//   * the boards stay in the XCD L2s for the whole sequence.  Block k of a part owns the same boards in every launch, and a queue
//     deals workgroup k of a dispatch to XCD (k + s) % 8 (tools/aql_probe.cpp, profiles/r05_c_*) -- with an s that nothing documents
//     as fixed: it was seen to move once, between this route's start-up and its first sequence, after the HIP runtime had started
//     its own queues (profiles/r05_g_*), and never again in 24 000 fuzzed sequences with queues being created and destroyed all the
//     while, in this process and in another (profiles/r05_o .. r05_q).  So only the first packet of
//     a chain acquires and only the last releases (system scope), the packets between carry the barrier bit and NO fence, and
//     the contract is checked instead of assumed: in the first launch every workgroup records the XCD it runs on (and the
//     boards it loads go into a snapshot buffer of the library); in every later launch every workgroup compares.  One mismatch
//     anywhere in the sequence and the call restores the snapshot and reports RERUN: the caller runs the same sequence through
//     the HIP stream route -- the result is the same bytes, a millisecond later.  The kernels are the resident variants of
//     step_kernels.hip (agent-scope board loads: a tile may change CUs inside its XCD from launch to launch; plain write-back
//     stores): per step only the action row comes in and the reward / done rows go out;
//   * argument blocks live in device memory the host writes through the PCIe BAR (one fence + read-back per segment): blocks
//     in host memory cost every wave a bus round trip (18 us per step, profiles/r05_a_*).
// Measured on 2^20 boards: 5.3-5.5 us per step at K = 200 (HIP: 8.0 as one chain, 7.0-7.1 as two), 6.3-6.5 at K = 20 (HIP 9.0-9.2);
// 65 536 boards 2.4-2.5 (HIP 3.3) -- profiles/r05_e_*.
//
// The call is SYNCHRONOUS on this route: it waits for the caller's stream to drain, runs the sequence and returns when the
// results are in memory (the HIP build of this image has no stream-wait-on-memory: hipExtMallocWithFlags(hipMallocSignalMemory)
// is refused).  Anything this route cannot do -- a stream being captured, batches whose boards do not fit the L2s, single steps
// (asynchronous by contract) -- takes the HIP stream route of step_kernels.hip unchanged.  Legal masks, observations and the
// log-normalised reward are served by the _extras variants of the resident kernels (the 32- or 64-byte observation rows leave
// write-through, so that they do not push the boards out of the L2s): 15.4 -> 13.1 us per 2^20-board step with f32
// observations, 12.5 -> 11.2 with bf16.
#pragma once
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <mutex>

#include "env2048.h"
#include "errors.hpp"
#include "helper_thread.hpp"
#include "tiling.hpp"

extern "C" const unsigned char env2048_direct_image[];  // the step kernels as a bare gfx950 code object (direct_image.S, build.py)
extern "C" const unsigned char env2048_direct_image_end[];

namespace env2048 {
namespace host {
namespace direct {

constexpr int NOT_TAKEN = 1;  // step_sequence(): nothing was enqueued, this call is not for the direct route (the caller takes the HIP route)
constexpr int RERUN = 2;      // step_sequence(): the sequence ran, a workgroup changed XCDs, the boards are back at their state before the call
constexpr int CHAINS = 2;
// Hardware queues are dealt to the four pipes of the compute engine in the order they are created, and two queues on ONE pipe do
// not run side by side (two chains on queues created four apart: 14-18 us per step instead of 5.4, profiles/r05_j_*).  The
// library therefore takes a whole round of four -- it uses the first two -- so that the HIP runtime's own queues, created before
// or after, keep the pipes they would have had without it: with only two, a process's third HIP stream landed on the pipe of its
// first (a 2-chain stream sequence next to the library's queues took 23.6 us per step); with four the same sequence takes 7.0 us
// with the library's queues and 7.25 without (tools/diag_pipe_sharing.py, profiles/r05_w_*; a process with more than four HIP
// streams of its own collides with itself either way: 11 us).
constexpr int QUEUES_CREATED = 4;
constexpr uint32_t QUEUE_PACKETS = 1024;
constexpr uint32_t SEGMENT_STEPS = 256;               // argument blocks are written (and flushed) a segment at a time
constexpr uint32_t RING_SEGMENTS = 4;                 // ... into a ring of this many segments per chain
constexpr size_t BLOCK_STRIDE = 128;                  // bytes per argument block (the kernel reads 120)
constexpr uint32_t MAX_BLOCKS_PER_PART = 2048;        // workgroups of one launch (2^21 boards as ONE small-tiling part would be 4096: never planned)
constexpr uint64_t RESIDENT_MAX_BOARDS = 3ull << 19;  // 1.5 x 2^20: boards + output rows must fit the eight 4 MiB L2s -- direct / stream us per step at
                                                      // 2^20: 5.8 / 7.4, 1.25 x 2^20: 8.3 / 9.0, 1.5 x: 8.8 / 9.7, 1.75 x: 12.0 / 11.1, 2^21: 12.9 / 12.2 (profiles/r05_z_*)
constexpr uint64_t TWO_CHAINS_MIN_BOARDS = 1ull << 18;  // below: one chain (65 536 boards: 2.43 vs 2.41 us; 2^18: 3.3 vs 2.6)
constexpr uint32_t DEFAULT_MIN_STEPS = 8;             // shorter sequences stay asynchronous launches on the caller's stream

// the arguments of env2048_step_resident_* as the code object lays them out
struct ResidentArgs {
    uint64_t boards, actions, n, env_base;
    uint32_t flags, seed_lo, seed_hi, step_lo, step_hi, pad0;
    uint64_t reward, done;
    uint32_t mode, pad1;  // RESIDENT_PRETEND_MOVED of step_kernels.hip (test hook), else 0
    uint64_t owner, placement_error, snapshot;
    uint64_t legal, obs;  // (read by the _extras kernels only)
};
static_assert(sizeof(ResidentArgs) == 120 && offsetof(ResidentArgs, reward) == 56 && offsetof(ResidentArgs, mode) == 72 &&
                  offsetof(ResidentArgs, owner) == 80 && offsetof(ResidentArgs, placement_error) == 88 && offsetof(ResidentArgs, snapshot) == 96 &&
                  offsetof(ResidentArgs, legal) == 104 && offsetof(ResidentArgs, obs) == 112,
              "argument layout of env2048_step_resident_* (step_kernels.hip)");
constexpr uint32_t MODE_PRETEND_MOVED = 2u;  // (test hook) the workgroups of this launch claim to run on another XCD

struct KernelRef {
    uint64_t object = 0;
    uint32_t group = 0, priv = 0, kernarg = 0;
};

inline double now_us()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e6 + (double)t.tv_nsec * 1e-3;
}

struct Device {
    bool ok = false;
    char reason[160] = "not initialised";
    std::mutex call_mutex;  // one direct sequence at a time per device
    hsa_agent_t gpu{}, cpu{};
    hsa_amd_memory_pool_t local_pool{}, host_pool{};
    hsa_queue_t *queue[QUEUES_CREATED] = {};  // [0, CHAINS) carry the chains, the rest only hold their pipes
    KernelRef resident[2][2][2][2];  // [legal / obs / normalised reward][small tiling][random actions][first launch of a chain]
    hsa_executable_t executable{};
    hsa_code_object_reader_t reader{};
    uint8_t *ring[CHAINS] = {};                        // argument blocks: device memory, host-visible through the BAR
    uint64_t next_block[CHAINS] = {};                  // blocks ever written per chain (slot = index % ring size)
    hsa_signal_t segment_signal[CHAINS][RING_SEGMENTS];  // completion of the last packet of the segment that last used this part of the ring
    bool segment_pending[CHAINS][RING_SEGMENTS] = {};
    volatile uint32_t *placement_error = nullptr;      // pinned host word
    uint32_t *owner[CHAINS] = {};                      // device: XCD of workgroup k in the first launch of the current sequence
    uint8_t *snapshot = nullptr;                       // device: the boards as the current sequence found them (hipMalloc, grown on demand)
    uint64_t snapshot_boards = 0;
    volatile uint32_t *hdp_flush = nullptr;
    uint64_t timestamp_hz = 0;
    bool timing = false;
    // counters (env2048_direct_info)
    uint64_t calls = 0, steps = 0, packets = 0, declined = 0, reruns = 0;
    uint64_t last_gpu_ns = 0;
    uint32_t last_chains = 0;
};

inline int set_reason(Device *d, const char *what, hsa_status_t st)
{
    const char *text = nullptr;
    if (st != HSA_STATUS_SUCCESS) hsa_status_string(st, &text);
    snprintf(d->reason, sizeof d->reason, "%s%s%s", what, text ? ": " : "", text ? text : "");
    return 0;
}

// ---- start-up -------------------------------------------------------------------------------------------------------------------
struct AgentSearch {
    uint32_t domain, bdf;
    hsa_agent_t gpu, cpu;
    bool have_gpu, have_cpu;
};
inline hsa_status_t visit_agent(hsa_agent_t agent, void *data)
{
    AgentSearch *s = static_cast<AgentSearch *>(data);
    hsa_device_type_t type;
    if (hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (type == HSA_DEVICE_TYPE_CPU && !s->have_cpu) {
        s->cpu = agent;
        s->have_cpu = true;
    } else if (type == HSA_DEVICE_TYPE_GPU && !s->have_gpu) {
        uint32_t bdf = 0, domain = 0;
        hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
        hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
        if (bdf == s->bdf && domain == s->domain) {
            s->gpu = agent;
            s->have_gpu = true;
        }
    }
    return HSA_STATUS_SUCCESS;
}
struct PoolSearch {
    hsa_amd_memory_pool_t pool;
    bool found;
    bool want_kernarg;  // host side: the fine-grained kernarg pool; device side: the coarse-grained local pool
};
inline hsa_status_t visit_pool(hsa_amd_memory_pool_t pool, void *data)
{
    PoolSearch *s = static_cast<PoolSearch *>(data);
    hsa_amd_segment_t segment;
    uint32_t flags = 0;
    bool alloc = false;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &segment);
    if (segment != HSA_AMD_SEGMENT_GLOBAL || s->found) return HSA_STATUS_SUCCESS;
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (!alloc) return HSA_STATUS_SUCCESS;
    if (s->want_kernarg ? (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) != 0 : (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) != 0) {
        s->pool = pool;
        s->found = true;
    }
    return HSA_STATUS_SUCCESS;
}

inline bool lookup(Device *d, const char *name, KernelRef *k)
{
    hsa_executable_symbol_t sym;
    if (hsa_executable_get_symbol_by_name(d->executable, name, &d->gpu, &sym) != HSA_STATUS_SUCCESS) return false;
    return hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k->object) == HSA_STATUS_SUCCESS &&
           hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k->group) == HSA_STATUS_SUCCESS &&
           hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k->priv) == HSA_STATUS_SUCCESS &&
           hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k->kernarg) == HSA_STATUS_SUCCESS;
}

inline uint16_t packet_header(bool acquire, bool release)
{
    return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                      ((acquire ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                      ((release ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

// One kernel-dispatch packet into `q` (the caller is the queue's only producer): body, then header + setup as ONE 32-bit release
// store, then the doorbell.  Waits while the queue is full -- bounded: every packet in it belongs to a kernel that terminates.
inline void push_packet(hsa_queue_t *q, const KernelRef &k, uint32_t blocks, uint64_t kernarg_address, uint16_t header, hsa_signal_t completion)
{
    const uint64_t index = hsa_queue_load_write_index_relaxed(q);
    while (index - hsa_queue_load_read_index_scacquire(q) >= q->size) _mm_pause();
    hsa_kernel_dispatch_packet_t *p = reinterpret_cast<hsa_kernel_dispatch_packet_t *>(q->base_address) + (index & (q->size - 1));
    p->workgroup_size_x = (uint16_t)BLOCK;
    p->workgroup_size_y = 1;
    p->workgroup_size_z = 1;
    p->reserved0 = 0;
    p->grid_size_x = blocks * (uint32_t)BLOCK;
    p->grid_size_y = 1;
    p->grid_size_z = 1;
    p->private_segment_size = k.priv;
    p->group_segment_size = k.group;
    p->kernel_object = k.object;
    p->kernarg_address = reinterpret_cast<void *>(kernarg_address);
    p->reserved2 = 0;
    p->completion_signal = completion;
    const uint32_t word = (uint32_t)header | ((uint32_t)(1u << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16);
    __atomic_store_n(reinterpret_cast<uint32_t *>(p), word, __ATOMIC_RELEASE);
    hsa_queue_store_write_index_relaxed(q, index + 1);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)index);
}

// Wait until `signal` drops below 1: poll for `spin_us`, then sleep in the runtime a millisecond at a time.  false after
// `limit_us` (a lost kernel must not hang the host).
inline bool wait_signal(hsa_signal_t signal, double spin_us, double limit_us)
{
    const double t0 = now_us();
    for (;;) {
        if (hsa_signal_load_scacquire(signal) < 1) return true;
        const double waited = now_us() - t0;
        if (waited > limit_us) return false;
        if (waited > spin_us) {
            if (hsa_signal_wait_scacquire(signal, HSA_SIGNAL_CONDITION_LT, 1, 1000000ull /* ns-ish ticks */, HSA_WAIT_STATE_BLOCKED) < 1) return true;
        } else {
            _mm_pause();
        }
    }
}

// The argument blocks written so far reach device memory: write-combining buffers out (sfence), host data path flushed, and a
// read of the last word behind them (a read does not pass the posted writes in front of it).
inline void flush_blocks(Device *d, const uint8_t *last_written)
{
    _mm_sfence();
    if (d->hdp_flush) *d->hdp_flush = 1u;
    (void)*reinterpret_cast<const volatile uint64_t *>(last_written);
}

inline Device *create(int device)
{
    Device *d = new Device;
    const char *env = getenv("ENV2048_DIRECT");
    if (env && env[0] == '0') {
        set_reason(d, "disabled by ENV2048_DIRECT=0", HSA_STATUS_SUCCESS);
        return d;
    }
    if (env2048_direct_image_end - env2048_direct_image < 64) {
        set_reason(d, "this build carries no direct-dispatch code object", HSA_STATUS_SUCCESS);
        return d;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        (void)hipGetLastError();
        set_reason(d, "hipGetDeviceProperties failed", HSA_STATUS_SUCCESS);
        return d;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(d->reason, sizeof d->reason, "device is %s, the code object is gfx950", prop.gcnArchName);
        return d;
    }
    hsa_status_t st = hsa_init();  // (the HIP runtime holds a reference already: this adds one)
    if (st != HSA_STATUS_SUCCESS) return set_reason(d, "hsa_init", st), d;
    AgentSearch agents{(uint32_t)prop.pciDomainID, ((uint32_t)prop.pciBusID << 8) | ((uint32_t)prop.pciDeviceID << 3), {}, {}, false, false};
    hsa_iterate_agents(visit_agent, &agents);
    if (!agents.have_gpu || !agents.have_cpu) return set_reason(d, "no ROCr agent with the PCI address of this HIP device", HSA_STATUS_SUCCESS), d;
    d->gpu = agents.gpu;
    d->cpu = agents.cpu;
    uint32_t num_xcc = 0;
    hsa_agent_get_info(d->gpu, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_NUM_XCC, &num_xcc);
    if (num_xcc != 8) {
        snprintf(d->reason, sizeof d->reason, "the device presents %u XCDs, the placement rule is for 8", num_xcc);
        return d;
    }
    PoolSearch local{{}, false, false}, host{{}, false, true};
    hsa_amd_agent_iterate_memory_pools(d->gpu, visit_pool, &local);
    hsa_amd_agent_iterate_memory_pools(d->cpu, visit_pool, &host);
    if (!local.found || !host.found) return set_reason(d, "memory pools not found", HSA_STATUS_SUCCESS), d;
    d->local_pool = local.pool;
    d->host_pool = host.pool;
    hsa_amd_memory_pool_access_t access = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
    hsa_amd_agent_memory_pool_get_info(d->cpu, d->local_pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &access);
    if (access == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED)
        return set_reason(d, "device memory is not host-visible (no large BAR): argument blocks cannot be written in place", HSA_STATUS_SUCCESS), d;
    // code object
    st = hsa_code_object_reader_create_from_memory(env2048_direct_image, (size_t)(env2048_direct_image_end - env2048_direct_image), &d->reader);
    if (st != HSA_STATUS_SUCCESS) return set_reason(d, "code object reader", st), d;
    hsa_profile_t profile = HSA_PROFILE_BASE;
    hsa_agent_get_info(d->gpu, HSA_AGENT_INFO_PROFILE, &profile);
    st = hsa_executable_create_alt(profile, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &d->executable);
    if (st == HSA_STATUS_SUCCESS) st = hsa_executable_load_agent_code_object(d->executable, d->gpu, d->reader, nullptr, nullptr);
    if (st == HSA_STATUS_SUCCESS) st = hsa_executable_freeze(d->executable, nullptr);
    if (st != HSA_STATUS_SUCCESS) return set_reason(d, "loading the code object", st), d;
    for (int x = 0; x < 2; ++x)
        for (int s = 0; s < 2; ++s)
            for (int r = 0; r < 2; ++r)
                for (int f = 0; f < 2; ++f) {
                    char name[96];
                    snprintf(name, sizeof name, "env2048_step_resident_%s%s%s%s.kd", s ? "small" : "large", x ? "_extras" : "", r ? "_random" : "",
                             f ? "_first" : "");
                    KernelRef &k = d->resident[x][s][r][f];
                    if (!lookup(d, name, &k) || k.kernarg != sizeof(ResidentArgs) || k.priv != 0) {
                        snprintf(d->reason, sizeof d->reason, "kernel %s missing or of another shape", name);
                        return d;
                    }
                }
    // queues, signals, memory
    for (int q = 0; q < QUEUES_CREATED; ++q) {
        st = hsa_queue_create(d->gpu, q < CHAINS ? QUEUE_PACKETS : 64u, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &d->queue[q]);
        if (st != HSA_STATUS_SUCCESS) return set_reason(d, "hsa_queue_create", st), d;
    }
    for (int c = 0; c < CHAINS; ++c) {
        for (uint32_t s = 0; s < RING_SEGMENTS; ++s)
            if ((st = hsa_signal_create(0, 0, nullptr, &d->segment_signal[c][s])) != HSA_STATUS_SUCCESS) return set_reason(d, "hsa_signal_create", st), d;
        void *ring = nullptr;
        st = hsa_amd_memory_pool_allocate(d->local_pool, (size_t)RING_SEGMENTS * SEGMENT_STEPS * BLOCK_STRIDE + MAX_BLOCKS_PER_PART * 4, 0, &ring);
        if (st == HSA_STATUS_SUCCESS) {
            hsa_agent_t both[2] = {d->cpu, d->gpu};
            st = hsa_amd_agents_allow_access(2, both, nullptr, ring);
        }
        if (st != HSA_STATUS_SUCCESS) return set_reason(d, "host-visible device memory for the argument blocks", st), d;
        d->ring[c] = static_cast<uint8_t *>(ring);
        d->owner[c] = reinterpret_cast<uint32_t *>(d->ring[c] + (size_t)RING_SEGMENTS * SEGMENT_STEPS * BLOCK_STRIDE);  // (behind the ring)
    }
    void *word = nullptr;
    st = hsa_amd_memory_pool_allocate(d->host_pool, 4096, 0, &word);
    if (st == HSA_STATUS_SUCCESS) st = hsa_amd_agents_allow_access(1, &d->gpu, nullptr, word);
    if (st != HSA_STATUS_SUCCESS) return set_reason(d, "pinned host word", st), d;
    memset(word, 0, 4096);
    d->placement_error = static_cast<volatile uint32_t *>(word);
    hsa_amd_hdp_flush_t hdp{};
    if (hsa_agent_get_info(d->gpu, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &hdp) == HSA_STATUS_SUCCESS) d->hdp_flush = hdp.HDP_MEM_FLUSH_CNTL;
    hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &d->timestamp_hz);
    const char *timing = getenv("ENV2048_DIRECT_TIMING");
    d->timing = !(timing && timing[0] == '0') && d->timestamp_hz != 0;
    if (d->timing)
        for (int c = 0; c < CHAINS; ++c)
            if (hsa_amd_profiling_set_profiler_enabled(d->queue[c], 1) != HSA_STATUS_SUCCESS) d->timing = false;
    d->ok = true;
    snprintf(d->reason, sizeof d->reason, "ready: %d queues, argument blocks in host-written device memory, timing %s", CHAINS, d->timing ? "on" : "off");
    return d;
}

inline PerDeviceRegistry<Device> &registry()
{
    static PerDeviceRegistry<Device> r;
    return r;
}
inline Device *device_for(int device)  // (the caller has selected `device`)
{
    return registry().get(device, [device] { return create(device); });
}

inline uint32_t min_steps()
{
    static const uint32_t v = [] {
        const char *e = getenv("ENV2048_DIRECT_MIN_STEPS");
        const long parsed = e ? atol(e) : 0;
        return parsed > 0 ? (uint32_t)parsed : DEFAULT_MIN_STEPS;
    }();
    return v;
}

// Is this call one for the direct route?  (pure: sizes and flags only -- env2048_direct_plan reports it; every output
// env2048_step_sequence can ask for is served: reward / done, and legal / obs / the normalised reward by the _extras kernels)
inline uint64_t max_boards()
{
    static const uint64_t v = [] {  // (ENV2048_EXP_DIRECT_MAX_LOG2: experiment switch, up to 2^23: profiles/r05_z_*)
        const char *e = getenv("ENV2048_EXP_DIRECT_MAX_LOG2");
        const long parsed = e ? atol(e) : 0;
        return parsed >= 10 && parsed <= 23 ? 1ull << parsed : RESIDENT_MAX_BOARDS;
    }();
    return v;
}
inline bool eligible(uint64_t n, uint32_t count, uint32_t flags)
{
    return !(flags & (ENV2048_FLAG_NO_DIRECT | ENV2048_FLAG_STORES_WRITE_THROUGH | ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE)) && n > 0 &&
           n <= max_boards() && count >= min_steps();
}
inline int chains_for(uint64_t n) { return n >= TWO_CHAINS_MIN_BOARDS ? CHAINS : 1; }

// How a batch of n boards is cut into the parts the chains step (pure; env2048_direct_plan reports it): one part below
// TWO_CHAINS_MIN_BOARDS, else two, the second starting on a block boundary of the large tiling; a part's tiling follows ITS size, as a
// launch of that size would.  false: a part would need more workgroups than the owner table holds (never for n <= RESIDENT_MAX_BOARDS).
struct Parts {
    int chains;
    uint64_t first[CHAINS + 1];  // part c = boards [first[c], first[c + 1])
    bool small[CHAINS];
    uint32_t blocks[CHAINS];
};
inline bool plan_parts(uint64_t n, Parts *p)
{
    p->chains = chains_for(n);
    const uint64_t half = p->chains == 2 ? (n / 2) / block_boards(BOARDS_PER_LANE) * block_boards(BOARDS_PER_LANE) : n;
    p->first[0] = 0;
    p->first[1] = p->chains == 2 ? half : n;
    p->first[p->chains] = n;
    for (int c = 0; c < p->chains; ++c) {
        const uint64_t n_part = p->first[c + 1] - p->first[c];
        p->small[c] = plan_small_tiling(n_part, 0);
        const int bpl = p->small[c] ? 1 : BOARDS_PER_LANE;
        p->blocks[c] = (uint32_t)((n_part + block_boards(bpl) - 1) / block_boards(bpl));
        if (p->blocks[c] > MAX_BLOCKS_PER_PART) return false;
    }
    return true;
}
// Steps of the segment that starts at step `first` of a `count`-step sequence: the first segment is ONE step (its argument blocks
// are flushed and its packets ring the doorbells before the rest of the blocks are written: the device starts 2-3 us earlier), the
// others up to SEGMENT_STEPS.
inline uint32_t segment_steps(uint32_t first, uint32_t count)
{
    return first == 0 ? 1u : (count - first < SEGMENT_STEPS ? count - first : SEGMENT_STEPS);
}

struct SequenceArgs {
    uint64_t n, env_base;
    uint8_t *boards;
    const uint8_t *const *actions;  // count row pointers, or NULL with ENV2048_FLAG_RANDOM_ACTIONS
    int32_t *reward;
    uint8_t *done;
    uint8_t *legal;  // optional
    void *obs;       // optional: 16 x 4 (or x 2 with ENV2048_FLAG_OBS_F16 / _BF16) bytes per board
    uint64_t row_stride;  // in boards
    uint64_t seed, step_index;
    uint32_t count, flags;
};

// Run the sequence on the library's queues.  ENV2048_OK; an error; NOT_TAKEN (nothing was enqueued: take the HIP route); or RERUN
// (the sequence ran, some workgroup found itself on another XCD than in the first launch, so what the chain computed cannot be
// trusted: `boards` hold their state from before the call again and the caller's stream is idle -- take the HIP route).
inline int step_sequence(int device, hipStream_t stream, const SequenceArgs &a)
{
    Device *d = device_for(device);
    if (!d || !d->ok) return NOT_TAKEN;
    std::lock_guard<std::mutex> call(d->call_mutex);
    if (!d->ok) return NOT_TAKEN;  // (switched off by the call this one waited for)
    static const bool timeline = getenv("ENV2048_DIRECT_DEBUG") != nullptr;  // (diagnostic: where the host's microseconds go)
    const double t_enter = timeline ? now_us() : 0.0;
    double t_stream = 0.0, t_first_packets = 0.0, t_all_packets = 0.0;
    uint64_t ticks_first_packets = 0, ticks_complete = 0;
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (stream && (hipStreamIsCapturing(stream, &capture) != hipSuccess || capture != hipStreamCaptureStatusNone)) {
        (void)hipGetLastError();
        ++d->declined;
        return NOT_TAKEN;  // (a capture records launches, it cannot record this)
    }
    // the snapshot the first launch fills (and a RERUN restores): grown on demand, never shrunk
    if (d->snapshot_boards < a.n) {
        if (d->snapshot) (void)hipFree(d->snapshot);
        d->snapshot = nullptr;
        d->snapshot_boards = 0;
        void *mem = nullptr;
        if (hipMalloc(&mem, 16 * a.n) != hipSuccess) {
            (void)hipGetLastError();
            ++d->declined;
            return NOT_TAKEN;
        }
        d->snapshot = static_cast<uint8_t *>(mem);
        d->snapshot_boards = a.n;
    }
    // everything already on the caller's stream comes first: the queues below know nothing of it
    hipError_t e = hipStreamQuery(stream);
    if (e == hipErrorNotReady) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "env2048_step_sequence: waiting for the caller's stream");

    if (timeline) t_stream = now_us();
    static const long pretend_moved_at = getenv("ENV2048_DIRECT_TEST_MOVE_AT") ? atol(getenv("ENV2048_DIRECT_TEST_MOVE_AT")) : -1;  // (test hook)
    const bool random_actions = (a.flags & ENV2048_FLAG_RANDOM_ACTIONS) != 0u;
    Parts parts;
    if (!plan_parts(a.n, &parts)) {
        ++d->declined;
        return NOT_TAKEN;
    }
    const uint64_t *const part_first = parts.first;
    const bool *const part_small = parts.small;
    const uint32_t *const part_blocks = parts.blocks;
    const int chains = parts.chains;
    const bool extras = a.legal || a.obs || (a.flags & ENV2048_FLAG_REWARD_LOG2_F32);
    const uint64_t obs_bytes_per_board = 16u * (obs_format(a.flags) == OBS_F32 ? 4u : 2u);
    uint32_t kernel_flags = a.flags & (ENV2048_FLAG_AUTO_RESET | ENV2048_FLAG_RANDOM_ACTIONS | ENV2048_FLAG_OBS_F16 | ENV2048_FLAG_OBS_BF16 |
                                       ENV2048_FLAG_REWARD_LOG2_F32);  // plain stores (the observation rows write-through: step_block)
    if (!misaligned(a.reward, 16) && !misaligned(a.done, 4) && a.row_stride % 4 == 0) kernel_flags |= INTERNAL_FLAG_WIDE_OUTPUTS;  // reward / done rows leave as 16-byte / 4-byte stores
    *d->placement_error = 0;
    const uint32_t ring_blocks = RING_SEGMENTS * SEGMENT_STEPS;
    const hsa_signal_t no_signal{0};
    bool timed_out = false;

    // (timing: what the first / last packet of each chain signals; the first packet's start is read before its signal -- the one of the
    // ring part its one-step segment occupies -- is given to a later segment of a long sequence)
    hsa_signal_t first_signal_used[CHAINS] = {}, last_signal_used[CHAINS] = {};
    uint32_t first_slot[CHAINS] = {};
    uint64_t first_start[CHAINS] = {};
    bool first_start_known[CHAINS] = {};
    auto note_first_start = [&](int c) {
        hsa_amd_profiling_dispatch_time_t t{};
        if (d->timing && !first_start_known[c] && first_signal_used[c].handle &&
            hsa_amd_profiling_get_dispatch_time(d->gpu, first_signal_used[c], &t) == HSA_STATUS_SUCCESS)
            first_start[c] = t.start;
        first_start_known[c] = true;
    };
    for (uint32_t first = 0, steps = 0; first < a.count && !timed_out; first += steps) {
        steps = segment_steps(first, a.count);
        const bool first_segment = first == 0, last_segment = first + steps == a.count;
        uint32_t segment_slot[CHAINS];
        // 1. the argument blocks of this segment (the part of the ring they go to is free once the segment that used it last is done)
        for (int c = 0; c < chains; ++c) {
            segment_slot[c] = (uint32_t)((d->next_block[c] / SEGMENT_STEPS) % RING_SEGMENTS);
            if (d->segment_pending[c][segment_slot[c]]) {
                if (!wait_signal(d->segment_signal[c][segment_slot[c]], 200.0, 30e6)) timed_out = true;
                d->segment_pending[c][segment_slot[c]] = false;
                if (!first_segment && segment_slot[c] == first_slot[c]) note_first_start(c);  // (before this slot's signal is rearmed below)
            }
            if (first_segment) first_slot[c] = segment_slot[c];
        }
        if (timed_out) break;
        for (int c = 0; c < chains; ++c) {
            const uint64_t off = part_first[c], n_part = part_first[c + 1] - off;
            uint8_t *last = nullptr;
            for (uint32_t t = 0; t < steps; ++t) {
                const uint64_t step = a.step_index + first + t, row = (uint64_t)(first + t) * a.row_stride;
                ResidentArgs r{};
                r.boards = (uint64_t)(uintptr_t)(a.boards + 16 * off);
                r.actions = random_actions || !a.actions ? 0 : (uint64_t)(uintptr_t)(a.actions[first + t] + off);
                r.n = n_part;
                r.env_base = a.env_base + off;
                r.flags = kernel_flags;
                r.seed_lo = (uint32_t)a.seed;
                r.seed_hi = (uint32_t)(a.seed >> 32);
                r.step_lo = (uint32_t)step;
                r.step_hi = (uint32_t)(step >> 32);
                r.reward = a.reward ? (uint64_t)(uintptr_t)(a.reward + row + off) : 0;
                r.done = (uint64_t)(uintptr_t)(a.done + row + off);
                r.mode = pretend_moved_at >= 0 && (long)(first + t) == pretend_moved_at ? MODE_PRETEND_MOVED : 0u;
                r.owner = (uint64_t)(uintptr_t)d->owner[c];
                r.placement_error = (uint64_t)(uintptr_t)d->placement_error;
                r.snapshot = (uint64_t)(uintptr_t)(d->snapshot + 16 * off);
                r.legal = a.legal ? (uint64_t)(uintptr_t)(a.legal + row + off) : 0;
                r.obs = a.obs ? (uint64_t)(uintptr_t)(static_cast<uint8_t *>(a.obs) + obs_bytes_per_board * (row + off)) : 0;
                last = d->ring[c] + ((d->next_block[c] + t) % ring_blocks) * BLOCK_STRIDE;
                memcpy(last, &r, sizeof r);
            }
            if (c + 1 == chains) flush_blocks(d, last);  // (one read-back orders every block written before it, the other chain's too)
        }
        // 2. the packets, step by step, chain by chain
        for (uint32_t t = 0; t < steps; ++t) {
            const bool first_packet = first_segment && t == 0, last_packet = last_segment && t + 1 == steps, segment_end = t + 1 == steps;
            for (int c = 0; c < chains; ++c) {
                const KernelRef &k = d->resident[extras ? 1 : 0][part_small[c] ? 1 : 0][random_actions ? 1 : 0][first_packet ? 1 : 0];
                hsa_signal_t completion = no_signal;
                if (segment_end) {
                    completion = d->segment_signal[c][segment_slot[c]];
                    hsa_signal_store_relaxed(completion, 1);
                    d->segment_pending[c][segment_slot[c]] = true;
                }
                if (first_packet) first_signal_used[c] = completion;  // (the first segment is one step long: its packet ends a segment)
                if (last_packet) last_signal_used[c] = completion;
                push_packet(d->queue[c], k, part_blocks[c], (uint64_t)(uintptr_t)(d->ring[c] + ((d->next_block[c] + t) % ring_blocks) * BLOCK_STRIDE),
                            packet_header(first_packet, last_packet), completion);
            }
        }
        if (timeline && first_segment) {
            t_first_packets = now_us();
            hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP, &ticks_first_packets);
        }
        for (int c = 0; c < chains; ++c) d->next_block[c] += SEGMENT_STEPS;  // (a segment always occupies one whole part of the ring)
        d->packets += (uint64_t)steps * chains;
    }
    if (timeline) t_all_packets = now_us();
    // 3. the results are in memory when the last packets have completed (their release fence is part of completing)
    for (int c = 0; c < chains; ++c)
        for (uint32_t s = 0; s < RING_SEGMENTS; ++s)
            if (d->segment_pending[c][s]) {
                if (!timed_out && !wait_signal(d->segment_signal[c][s], 50.0 + 8.0 * a.count, 30e6 + 100.0 * a.count)) timed_out = true;
                d->segment_pending[c][s] = false;
            }
    if (timed_out) {
        d->ok = false;
        set_reason(d, "a direct sequence did not complete in time: route switched off", HSA_STATUS_SUCCESS);
        return fail(ENV2048_ERR_DEVICE, "env2048_step_sequence: the direct-dispatch sequence did not complete (route switched off for this device)");
    }
    if (timeline) hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP, &ticks_complete);
    if (timeline)
        fprintf(stderr, "[direct] K = %u: stream idle after %.2f us, first packets out after %.2f, all packets after %.2f, complete after %.2f\n", a.count,
                t_stream - t_enter, t_first_packets - t_enter, t_all_packets - t_enter, now_us() - t_enter);
    if (d->timing) {  // device-side duration of the call: first packet's start to last packet's end, the longer of the chains
        static const bool debug = getenv("ENV2048_DIRECT_DEBUG") != nullptr;
        uint64_t longest = 0;
        for (int c = 0; c < chains; ++c) {
            hsa_amd_profiling_dispatch_time_t t0{}, t1{};
            note_first_start(c);
            t0.start = first_start[c];
            const hsa_status_t s0 = first_start[c] ? HSA_STATUS_SUCCESS : HSA_STATUS_ERROR;
            const hsa_status_t s1 = hsa_amd_profiling_get_dispatch_time(d->gpu, last_signal_used[c], &t1);
            if (debug)
                fprintf(stderr, "[direct] chain %d: first kernel starts %.2f us after its doorbell, the host sees completion %.2f us after the last kernel ended\n", c,
                        ((double)t0.start - (double)ticks_first_packets) * 1e6 / (double)d->timestamp_hz,
                        ((double)ticks_complete - (double)t1.end) * 1e6 / (double)d->timestamp_hz);
            if (s0 == HSA_STATUS_SUCCESS && s1 == HSA_STATUS_SUCCESS && t0.start != 0 && t1.end > t0.start && t1.end - t0.start < d->timestamp_hz * 3600ull)
                longest = t1.end - t0.start > longest ? t1.end - t0.start : longest;
        }
        d->last_gpu_ns = (uint64_t)((double)longest * 1e9 / (double)d->timestamp_hz);
    }
    ++d->calls;
    d->steps += a.count;
    d->last_chains = (uint32_t)chains;
    if (*d->placement_error != 0) {
        // A workgroup ran on another XCD than the one whose L2 holds its tile: whatever the chain computed from then on may rest on stale boards.  Back to the snapshot the first
        // launch took; the caller runs the sequence again as launches on its stream.
        const uint32_t word = *d->placement_error;
        ++d->reruns;
        snprintf(d->reason, sizeof d->reason, "ready (last rerun: workgroup %u of a launch ran on XCD %u, its tile's home is XCD %u)", (word & 0xffffu) - 1u,
                 (word >> 16) & 0xffu, word >> 24);
        e = hipMemcpyAsync(a.boards, d->snapshot, 16 * a.n, hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return hip_fail(e, "env2048_step_sequence: restoring the boards after a direct-dispatch sequence that has to be rerun");
        return RERUN;
    }
    return ENV2048_OK;
}

}  // namespace direct
}  // namespace host
}  // namespace env2048
