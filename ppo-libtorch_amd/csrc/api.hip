// ppo-libtorch_amd/csrc/api.hip -- the C-ABI of include/ppo_hip.h: context, buffers, launch sequencing, RCCL.
//
// Host language is C++ because the reference is C++; nothing here depends on LibTorch.  The context owns
// every device buffer (allocated once in ppo_ctx_create) and one HIP stream; every entry point only enqueues work.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "generic.hpp"
#include "ppo_internal.hpp"

// ---------------------------------------------------------------------------------------------------------
// RCCL, bound lazily (single-GPU use never loads it).  One all-reduce per optimizer step.
// ---------------------------------------------------------------------------------------------------------
namespace rccl {
typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(Comm*, int, UniqueId, int);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroy_t)(Comm);
typedef const char* (*GetErrorString_t)(int);
static GetUniqueId_t GetUniqueId;
static CommInitRank_t CommInitRank;
static AllReduce_t AllReduce;
static CommDestroy_t CommDestroy;
static GetErrorString_t GetErrorString;
static const int kFloat32 = 7, kFloat64 = 8, kSum = 0;
static bool load(std::string& err) {
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (AllReduce) return true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // reuse the copy a host process (e.g. PyTorch) already mapped
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    GetUniqueId = (GetUniqueId_t)dlsym(h, "ncclGetUniqueId");
    CommInitRank = (CommInitRank_t)dlsym(h, "ncclCommInitRank");
    AllReduce = (AllReduce_t)dlsym(h, "ncclAllReduce");
    CommDestroy = (CommDestroy_t)dlsym(h, "ncclCommDestroy");
    GetErrorString = (GetErrorString_t)dlsym(h, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) { err = "librccl lacks nccl* symbols"; AllReduce = nullptr; return false; }
    return true;
}
}  // namespace rccl

// ---------------------------------------------------------------------------------------------------------
// In-process communicator: several contexts of ONE process (one host thread each) on one or more GPUs.  Same semantics as
// the RCCL path -- every rank calls the all-reduce, the sum is formed in rank order (deterministic) -- without any RCCL:
// the last rank to arrive runs one kernel on its own stream that reads every rank's buffer and writes the sum back to all.
// ---------------------------------------------------------------------------------------------------------
struct LocalGroup {
    int n = 0;
    int device = -1;              // every member lives on this device (ppo_comm_init_local rejects others)
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, joined = 0;
    uint64_t generation = 0;
    bool failed = false;          // a member's launch failed inside the rendezvous: every waiter returns PPO_ERR_COMM instead of hanging
    void* bufs[8] = {};
    hipEvent_t ready[8] = {};
    hipEvent_t done[2] = { nullptr, nullptr };
    ~LocalGroup() { for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e); }
};
// ---------------------------------------------------------------------------------------------------------
// One-shot direct exchange: the all-reduce of a latency-bound payload (36.6 KB of gradient per optimizer step) WITHOUT a
// ring.  Every rank owns a small exchange buffer in fine-grained device memory -- two slots of payload + a flag each -- that its peers
// map through HIP IPC handles (one process per GPU; over xGMI each peer is one point-to-point hop).  An all-reduce is ONE kernel per rank:
// publish the own payload and flag, then for every rank in rank order wait for its flag and add its payload -- a fixed order, so every rank
// forms bit-identical sums -- straight out of the peer's memory.  Two slots suffice: a rank overwrites slot s two calls later, and it cannot
// get there before every peer has finished reading the call in between (whose sum it needed to proceed).
// RCCL stays available (ppo_comm_init); this path is chosen by ppo_comm_init_exchange.
// ---------------------------------------------------------------------------------------------------------
struct ExchangeComm {
    void* own = nullptr;              // [2 parities][8 source ranks][slot_bytes] payload slots the PEERS write into, then flags and counters
                                      // (fine-grained device memory of this rank; layout: kernels_update.hip, xchg_slot)
    size_t slot_bytes = 0;
    void* peer[8] = {};               // mapped exchange buffers of every rank (peer[rank] == own)
    bool opened[8] = {};
    uint64_t seq = 0;                 // calls so far (same on every rank)
    int32_t* timeout_flag = nullptr;  // device int32[4]: [0] set by a kernel whose bounded wait for a peer ran out; [2..3] = the wait limit as a u64 in
                                      // ticks of the 100 MHz counter (kernels_update.hip: xchg_wait_all)
    double wait_seconds = 30.0;       // larger than any realistic host-side skew between ranks (checkpoint write, statistics read-back, code-object load)
};
struct XchgPtrs { void* p[8]; };
hipError_t launch_exchange_allreduce(void* buf, size_t count, bool f64, const XchgPtrs& peers, int rank, int n, size_t slot_bytes, uint64_t seq,
                                     int32_t* timeout_flag, hipStream_t s);
static std::map<int64_t, std::shared_ptr<LocalGroup>> g_local_groups;
static std::mutex g_local_groups_mu;
struct PtrPack { void* p[8]; };
hipError_t launch_local_allreduce(const PtrPack& pk, int n, size_t count, bool f64, hipStream_t s);

// pinned host block a statistics snapshot lands in (device pieces first, then the host-side training state at snapshot time)
struct StatsSnap {
    StepStats st;
    double cf[2];
    double ev[PPO_EV_BLOCKS * 4];
    EpisodeRing ring;
    double gstats[PPO_GSTAT_DOUBLES];
    int32_t error_flag, xchg_flag;
    // host side
    bool have_step, have_ev, have_gstats;
    int world;
    int64_t B, global_step, opt_step, updates;
    double lr;
};

struct ppo_ctx {
    ppo_config cfg{};
    NetLayout L{};
    LossParams hp{};
    hipStream_t stream = nullptr;
    // generic networks in bf16 storage: the critic's forward and backward passes of a minibatch step run on a second stream beside the actor's
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_gather = nullptr;
    // ppo_update tells a minibatch step which rows the NEXT step will gather (nullptr: none): the step requests that gather on the second stream once its
    // own backward passes are done, beside its reduction / optimizer tail; gen_pre_* = what has been gathered ahead
    const int32_t* gen_next_idx = nullptr; int64_t gen_next_M = 0;
    const int32_t* gen_pre_idx = nullptr; int64_t gen_pre_M = 0;
    std::string err;
    int T = 0, N = 0, O = 0, H = 0, A = 0;
    int64_t B = 0, MB = 0;
    int n_mb = 0, steps_per_update = 0;
    int world = 1, rank = 0;
    rccl::Comm comm = nullptr;
    std::shared_ptr<LocalGroup> lgroup;   // in-process communicator (exclusive with comm)
    hipEvent_t lg_ready = nullptr;
    std::unique_ptr<ExchangeComm> xchg;   // one-shot direct exchange over IPC peer buffers (exclusive with comm / lgroup)

    std::vector<void*> allocs;
    void* buf[PPO_BUF_COUNT_] = {};
    size_t buf_bytes[PPO_BUF_COUNT_] = {};

    float* reset_table = nullptr;
    int reset_cap = 0;
    int32_t* error_flag = nullptr;
    float* slab = nullptr;
    double* stat_slab = nullptr;
    double* loss_sums = nullptr;
    float4* adv_norm = nullptr;         // [steps_per_update + 1] { mean, 1 / (std + 1e-8), std, 0 } per minibatch slot, from adv_stats once per update
    AdvStat* adv_stats = nullptr;       // [steps_per_update] + 1 scratch slot; sits right BEHIND gstats so that one all-reduce per update carries both
    double* gstats = nullptr;           // [PPO_GSTAT_DOUBLES] job-global statistics block of a sharded run (ppo_internal.hpp: PPO_GSTAT_*)
    bool have_gstats = false;           // the block holds the all-reduced statistics of the last ppo_update
    AdamCoef* adam_coefs = nullptr;     // device [steps_per_update + 1]
    AdamCoef* adam_coefs_h = nullptr;   // pinned mirror, two halves used alternately
    hipEvent_t coef_copied[2] = { nullptr, nullptr };  // H2D copy of each half has completed
    int coef_half = 0;
    StepStats* step_stats = nullptr;    // device [steps_per_update + 1]
    double* clipfrac_accum = nullptr;   // {sum, count}
    double* norm2 = nullptr;            // [12] per-tensor squared gradient norms of the current step
    double* ev_sums = nullptr;          // [PPO_EV_BLOCKS][4]
    float* rec_critic = nullptr;        // [B][16] packed sample records the matrix-core update kernel gathers from (launch_pack_records): critic | actor per sample
    float* rec_actor = nullptr;         // = rec_critic + 8 (not an allocation of its own)
    int32_t* row_counts = nullptr;      // [T]
    uint64_t* group_bits = nullptr;     // [T, ceil(N/64)] ballots of finished episodes
    EpisodeRing* ring = nullptr;
    float* scratch_obs = nullptr;       // [N,O] staging for AoS<->SoA conversions
    int max_blocks_per_net = 0;
    bool use_mfma = true;
    bool update_single_wave = false;   // PPO_KERNEL_UPDATE_ONE_WAVE
    bool rollout_vector = false;       // PPO_KERNEL_ROLLOUT_VECTOR
    // fp16 range of the matrix-core kernels (ppo_internal.hpp: weight_range_kernel): maxima of |parameter| by class, taken once per update, on the device and mirrored into pinned
    // host memory that the dispatch reads WITHOUT synchronising; wrange_dirty = the host wrote parameters since they were last computed from scratch
    uint32_t* wr_dev = nullptr;
    uint32_t* wr_host = nullptr;       // hipHostMalloc'ed, mapped: wr_host_dev is the device's address of the same words
    uint32_t* wr_host_dev = nullptr;
    hipStream_t wr_stream = nullptr;   // the once-per-update sweep runs here, beside the update's first launches (sweep_weight_range)
    bool wrange_dirty = true;
    float wr_cache[3] = { 0.0f, 0.0f, 0.0f };   // the mirror as last read: the pinned words are uncached for the host (~0.3 us a read), so they are read once per
    bool gen_opt_fused_ok = false;             // generic bf16 path: the last backward pass left the sums of squares gen_opt_fused needs, and nothing touched the gradient since
    bool gen_obs_bf_valid = false;             // generic bf16 path: GenericCtx::obs_bf holds THIS update's observations (set by the update's first step)
    bool wr_in_update = false;                 // rollout / update / stand-alone step, not per launch (an update moves a weight by less than 40 lr: the thresholds' margin)
    int64_t vector_fallback_launches = 0;   // launches that took a vector kernel because a weight did not fit fp16 (ppo_profile.vector_fallback_launches)
    unsigned long long* stamps = nullptr;  // [2][12] phase cycles of the diagnostic kernel variant
    GenericCtx* gen = nullptr;       // non-null: synthetic env / network other than 2 x 64 (generic.hpp); every L-dependent entry point dispatches on it
    uint8_t* cur_mask = nullptr;     // generic path: action mask of the observation in NEXT_OBS, [N, A]
    bool force_collectives = false;  // PPO_COMM_SELFTEST: world == 1 but the multi-rank path (RCCL included) is taken
    double* fused_partial = nullptr; // [fused_opt_blocks][12] per-workgroup sums of squares of the gradient
    int last_n_blocks[2] = { 0, 0 };
    int prof_every = 1;              // mode 2: bracket one update-kernel launch in prof_every (an event pair costs the stream ~3 us)
    int64_t prof_count = 0;
    bool prof_last_sampled = false;   // the last update launch was bracketed: so is the all-reduce that follows it
    bool stamping = false;           // diagnostic flavour of the update kernel (in-kernel phase stamps)

    // host-side training state
    double lr = 0.0;
    int64_t opt_step = 0;
    int64_t global_step = 0;
    int64_t updates = 0;
    int64_t num_updates_total = 0;
    int64_t rollout_steps = 0;      // sampling-stream position (rollout steps taken so far)
    int64_t act_calls = 0;
    bool fin_pending = false;
    bool have_ev = false;
    int last_stat_slot = -1;
    double last_global_M = 1.0;

    // statistics snapshot (ppo_stats_snapshot / ppo_stats_snapshot_read): everything ppo_read_stats decodes, copied asynchronously into ONE pinned block
    // behind the work enqueued so far; the reader waits for the snapshot's event only, so a host may enqueue the next iteration before it reads
    struct StatsSnap* snap = nullptr;   // [2]: a host that runs one iteration ahead takes snapshot k + 1 before it reads snapshot k
    hipEvent_t snap_ev[2] = { nullptr, nullptr };
    int snap_oldest = 0, snap_count = 0; // FIFO of pending snapshots

    // profiling: (start, stop) event pairs per instrumented launch
    uint32_t profiling = 0;         // bit k set: time launches of kind k
    struct Span { hipEvent_t a, b; int kind; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
};

enum { PROF_FWD_BWD = 0, PROF_GAE, PROF_ROLLOUT, PROF_OPT, PROF_REDUCE, PROF_ALLREDUCE, PROF_KINDS_ };

struct ProfScope {
    ppo_ctx* c;
    hipEvent_t a = nullptr, b = nullptr;
    int kind;
    static hipEvent_t get(ppo_ctx* c) {
        if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    ProfScope(ppo_ctx* ctx, int k, bool sampled_in = true) : c(ctx), kind(k) {
        if (!(c->profiling & (1u << kind)) || !sampled_in) return;
        a = get(c); b = get(c);
        if (a) (void)hipEventRecord(a, c->stream);
    }
    ~ProfScope() {
        if (!a || !b) return;
        (void)hipEventRecord(b, c->stream);
        c->spans.push_back({ a, b, kind });
    }
};

static thread_local std::string g_create_error;

// Every entry point that allocates, creates events or launches does so on the CONTEXT's device and leaves the caller's current device as it
// found it (a host that drives several contexts from one thread, or shares the thread with another HIP user such as PyTorch).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const ppo_ctx* c) {
        if (!c) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != c->cfg.device) switched = hipSetDevice(c->cfg.device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

static ppo_status fail(ppo_ctx* ctx, ppo_status code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                                                   \
    do {                                                                                                                    \
        hipError_t e_ = (call);                                                                                             \
        if (e_ != hipSuccess) return fail(ctx, PPO_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define NEED(ctx, cond, msg)                                        \
    do {                                                            \
        if (!(cond)) return fail(ctx, PPO_ERR_INVALID, "%s", msg);  \
    } while (0)

template <class Tp>
static hipError_t dalloc(ppo_ctx* c, Tp** p, size_t count, bool zero = true) {
    void* q = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(Tp), 16);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return e;
    c->allocs.push_back(q);
    if (zero) { e = hipMemset(q, 0, bytes); if (e != hipSuccess) return e; }
    *p = static_cast<Tp*>(q);
    return hipSuccess;
}
template <class Tp>
static hipError_t dalloc_buf(ppo_ctx* c, int which, size_t count) {
    Tp* p = nullptr;
    hipError_t e = dalloc(c, &p, count);
    c->buf[which] = p;
    c->buf_bytes[which] = count * sizeof(Tp);
    return e;
}
template <class Tp>
static Tp* B_(ppo_ctx* c, int which) { return static_cast<Tp*>(c->buf[which]); }

// ---------------------------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------------------------
namespace {
struct Mt19937 {
    uint32_t mt[624];
    int idx;
    explicit Mt19937(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int i = 0; i < 624; i++) {
                const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
};
}  // namespace

// std::mt19937(seed) + std::uniform_real_distribution<float>(-0.05f, 0.05f): reference CartPole.h:28-29, CartPole.cpp:3-4,96-100.
// libstdc++'s generate_canonical<float,24> consumes one 32-bit draw: float(u)/2^32, clamped below 1; then *(b-a)+a.
extern "C" ppo_status ppo_cartpole_reset_stream_h(int64_t seed, int64_t n_resets, float* out_h) {
    if (!out_h || n_resets < 0) return PPO_ERR_INVALID;
    Mt19937 g((uint32_t)seed);
    const float a = -0.05f, b = 0.05f;
    for (int64_t i = 0; i < n_resets * 4; i++) {
        float r = (float)g.next() / 4294967296.0f;
        if (r >= 1.0f) r = std::nextafter(1.0f, 0.0f);
        out_h[i] = r * (b - a) + a;
    }
    return PPO_OK;
}

static ppo_status ensure_reset_table(ppo_ctx* c, int64_t need) {
    if (c->cfg.env_kind != PPO_ENV_CARTPOLE) return PPO_OK;
    if (need <= c->reset_cap) return PPO_OK;
    int64_t cap = std::max<int64_t>(c->reset_cap, 1024);
    while (cap < need) cap *= 2;
    NEED(c, cap < (1ll << 28), "reset table would exceed 2^28 entries");
    std::vector<float> h((size_t)cap * 4);
    ppo_cartpole_reset_stream_h(c->cfg.seed, cap, h.data());
    float* d = nullptr;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), (size_t)cap * 4 * sizeof(float)));
    HIPCHK(c, hipMemcpy(d, h.data(), (size_t)cap * 4 * sizeof(float), hipMemcpyHostToDevice));
    if (c->reset_table) {
        c->allocs.erase(std::remove(c->allocs.begin(), c->allocs.end(), (void*)c->reset_table), c->allocs.end());
        (void)hipFree(c->reset_table);
    }
    c->allocs.push_back(d);
    c->reset_table = d;
    c->reset_cap = (int)cap;
    return PPO_OK;
}

// Direct exchange: a kernel whose bounded wait for a peer ran out has summed only the shares that arrived and marked the communicator dead.  The
// stream is idle here (every caller has just synchronised it): turn the flag into an error instead of letting diverged replicas train on.
static ppo_status comm_health(ppo_ctx* c) {
    if (!c->xchg || !c->xchg->timeout_flag) return PPO_OK;
    int32_t f = 0;
    HIPCHK(c, hipMemcpy(&f, c->xchg->timeout_flag, sizeof f, hipMemcpyDeviceToHost));
    if (f != 0)
        return fail(c, PPO_ERR_COMM, "direct exchange: an all-reduce gave up waiting for a peer after %.1f s; its sums were incomplete, the replicas have "
                                     "diverged and the communicator is dead (every later all-reduce returns at once)", c->xchg->wait_seconds);
    return PPO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------------------------------------
extern "C" int32_t ppo_abi_version(void) { return PPO_ABI_VERSION; }

extern "C" const char* ppo_last_error(const ppo_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" void ppo_ctx_destroy(ppo_ctx* c) {
    if (!c) return;
    // teardown is best effort: nothing useful can be done with a failing release, and there is no caller to tell
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && rccl::CommDestroy) rccl::CommDestroy(c->comm);
    if (c->lg_ready) (void)hipEventDestroy(c->lg_ready);
    if (c->xchg) {
        for (int r = 0; r < 8; r++) if (c->xchg->opened[r]) (void)hipIpcCloseMemHandle(c->xchg->peer[r]);
        if (c->xchg->own) (void)hipFree(c->xchg->own);
    }
    for (auto& sp : c->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->gen) { delete c->gen; c->gen = nullptr; }
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->adam_coefs_h) (void)hipHostFree(c->adam_coefs_h);
    if (c->wr_host) (void)hipHostFree(c->wr_host);
    if (c->snap) (void)hipHostFree(c->snap);
    for (hipEvent_t e : c->snap_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->coef_copied) if (e) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_gather) (void)hipEventDestroy(c->ev_gather);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->wr_stream) { (void)hipStreamSynchronize(c->wr_stream); (void)hipStreamDestroy(c->wr_stream); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" ppo_status ppo_ctx_create(const ppo_config* cfg, ppo_ctx** out) {
    if (!cfg || !out) return fail(nullptr, PPO_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(ppo_config)) return fail(nullptr, PPO_ERR_INVALID, "ppo_config.struct_size %d != %zu", cfg->struct_size, sizeof(ppo_config));
    const bool generic = cfg->env_kind == PPO_ENV_SYNTHETIC;
    if (!generic && (cfg->hidden != PPO_HIDDEN || cfg->n_hidden != 2))
        return fail(nullptr, PPO_ERR_UNSUPPORTED, "only the reference architecture (2 hidden layers of 64, Agent.cpp:25-59) is built for the reference's "
                    "environments; got %d x %d (other shapes run with env_kind = PPO_ENV_SYNTHETIC)", cfg->n_hidden, cfg->hidden);
    if (generic && (cfg->hidden < 1 || cfg->hidden > 2048 || cfg->n_hidden < 1 || cfg->n_hidden > GEN_MAX_LAYERS - 1 || cfg->obs_size < 1 || cfg->obs_size > 8192))
        return fail(nullptr, PPO_ERR_UNSUPPORTED, "generic network out of range: hidden %d (1..2048), n_hidden %d (1..%d), obs %d (1..8192)", cfg->hidden,
                    cfg->n_hidden, GEN_MAX_LAYERS - 1, cfg->obs_size);
    if (cfg->env_kind != PPO_ENV_CARTPOLE && cfg->env_kind != PPO_ENV_MOUNTAINCAR && cfg->env_kind != PPO_ENV_SYNTHETIC)
        return fail(nullptr, PPO_ERR_INVALID, "unknown env_kind %d", cfg->env_kind);
    if (cfg->dist_kind != PPO_DIST_CATEGORICAL && cfg->dist_kind != PPO_DIST_MASKED) return fail(nullptr, PPO_ERR_INVALID, "unknown dist_kind %d", cfg->dist_kind);
    if (cfg->compute_dtype != PPO_DTYPE_F32 && cfg->compute_dtype != PPO_DTYPE_BF16) return fail(nullptr, PPO_ERR_INVALID, "unknown compute_dtype %d", cfg->compute_dtype);
    if (cfg->kernel_flags & ~(PPO_KERNEL_ROLLOUT_VECTOR | PPO_KERNEL_UPDATE_VECTOR | PPO_KERNEL_UPDATE_ONE_WAVE | PPO_KERNEL_COMM_SELFTEST | PPO_KERNEL_GENERIC_CLASSIC | PPO_KERNEL_GENERIC_SPLIT_HEAD)) return fail(nullptr, PPO_ERR_INVALID, "unknown bits in kernel_flags 0x%x", cfg->kernel_flags);
    if (cfg->compute_dtype == PPO_DTYPE_BF16 && !generic)
        return fail(nullptr, PPO_ERR_UNSUPPORTED, "compute_dtype = PPO_DTYPE_BF16 applies to networks whose layers are GEMMs (env_kind = PPO_ENV_SYNTHETIC); the reference's "
                    "2 x 64 networks always compute in f32");
    const int env_obs = cfg->env_kind == PPO_ENV_CARTPOLE ? 4 : (cfg->env_kind == PPO_ENV_MOUNTAINCAR ? 2 : cfg->obs_size);
    if (cfg->obs_size != env_obs) {
        // the reference's runtime check in initEnvs (PPO_Discrete.cpp:370-375), same wording
        return fail(nullptr, PPO_ERR_INVALID,
                    "The environment returned an observation of size %d, but your config defined the expected observation size to be %d.\n"
                    "Have you properly defined your PPOConfig.toml file for your environment?", env_obs, cfg->obs_size);
    }
    if (cfg->n_heads < 1 || cfg->n_heads > PPO_MAX_HEADS) return fail(nullptr, PPO_ERR_INVALID, "n_heads must be in [1,%d]", PPO_MAX_HEADS);
    int A = 0;
    for (int h = 0; h < cfg->n_heads; h++) {
        if (cfg->head_dims[h] < 1) return fail(nullptr, PPO_ERR_INVALID, "head_dims[%d] < 1", h);
        A += cfg->head_dims[h];
    }
    if (A > PPO_MAX_ACT) return fail(nullptr, PPO_ERR_UNSUPPORTED, "sum(head_dims) = %d exceeds %d", A, PPO_MAX_ACT);
    if (cfg->num_envs < 1 || cfg->num_steps < 1 || cfg->num_minibatches < 1 || cfg->update_epochs < 1)
        return fail(nullptr, PPO_ERR_INVALID, "num_envs, num_steps, num_minibatches, update_epochs must be >= 1");
    const int64_t B = (int64_t)cfg->num_envs * cfg->num_steps;
    if (B / cfg->num_minibatches < 1) return fail(nullptr, PPO_ERR_INVALID, "minibatch_size = batch/num_minibatches is 0");
    if (B >= (1ll << 31)) return fail(nullptr, PPO_ERR_UNSUPPORTED, "batch of %lld rows exceeds int32 indexing", (long long)B);

    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore_device{ caller_device };
    hipError_t e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return fail(nullptr, PPO_ERR_HIP, "hipSetDevice(%d): %s", cfg->device, hipGetErrorString(e));
    ppo_ctx* c = new ppo_ctx();
    c->cfg = *cfg;
    if (c->cfg.global_num_envs <= 0) c->cfg.global_num_envs = cfg->num_envs;
    c->L = make_layout(cfg->obs_size, cfg->n_heads, cfg->head_dims);
    if (generic) {
        c->gen = new GenericCtx();
        c->gen->L = make_gen_layout(cfg->obs_size, cfg->hidden, cfg->n_hidden, cfg->n_heads, cfg->head_dims);
        // the shared code sizes the flat parameter / gradient / moment buffers from L.P; the 2 x 64 offsets in L are not used
        c->L.P = c->gen->L.P;
        c->L.n_tensors = c->gen->L.n_tensors;
        c->L.net_size[0] = c->L.net_size[1] = 1;   // no per-workgroup gradient slabs on this path
    }
    c->hp = LossParams{ cfg->clip_coef, cfg->ent_coef, cfg->vf_coef, cfg->norm_adv, cfg->clip_vloss, cfg->dist_kind };
    c->T = cfg->num_steps; c->N = cfg->num_envs; c->O = cfg->obs_size; c->H = cfg->n_heads; c->A = A;
    c->B = B;
    c->MB = B / cfg->num_minibatches;                 // int division, PPO_Discrete.cpp:247
    c->n_mb = (int)((B + c->MB - 1) / c->MB);          // a ragged tail is an extra short minibatch (:573-576)
    c->steps_per_update = cfg->update_epochs * c->n_mb;
    c->lr = (double)cfg->learning_rate;                // AdamWOptions(m_learning_rate): double(float lr), :76-78
    const int64_t global_B = c->cfg.global_num_envs * (int64_t)cfg->num_steps;
    c->num_updates_total = cfg->total_timesteps > 0 ? cfg->total_timesteps / global_B : 0;  // :496

#define CK(call)                                                                                         \
    do {                                                                                                 \
        hipError_t e2 = (call);                                                                          \
        if (e2 != hipSuccess) {                                                                          \
            fail(nullptr, PPO_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e2));                    \
            ppo_ctx_destroy(c);                                                                          \
            return PPO_ERR_HIP;                                                                          \
        }                                                                                                \
    } while (0)
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t TN = (size_t)c->T * c->N, N = c->N, P = c->L.P;
    CK(dalloc_buf<float>(c, PPO_BUF_OBS, TN * c->O));
    CK(dalloc_buf<int32_t>(c, PPO_BUF_ACTIONS, TN * c->H));
    CK(dalloc_buf<float>(c, PPO_BUF_LOGPROBS, TN));
    CK(dalloc_buf<float>(c, PPO_BUF_REWARDS, TN));
    CK(dalloc_buf<float>(c, PPO_BUF_DONES, TN));
    CK(dalloc_buf<float>(c, PPO_BUF_VALUES, TN));
    CK(dalloc_buf<uint8_t>(c, PPO_BUF_MASKS, TN * c->A));
    CK(dalloc_buf<float>(c, PPO_BUF_ADVANTAGES, TN));
    CK(dalloc_buf<float>(c, PPO_BUF_RETURNS, TN));
    CK(dalloc_buf<float>(c, PPO_BUF_NEXT_OBS, N * c->O));
    CK(dalloc_buf<int32_t>(c, PPO_BUF_NEXT_DONE, N));
    CK(dalloc_buf<float>(c, PPO_BUF_NEXT_VALUE, N));
    CK(dalloc_buf<float>(c, PPO_BUF_PARAMS, P));
    CK(dalloc_buf<float>(c, PPO_BUF_GRADS, P + 8));   // + loss sums that ride through the all-reduce when sharded
    c->buf_bytes[PPO_BUF_GRADS] = P * sizeof(float);
    CK(dalloc_buf<float>(c, PPO_BUF_EXP_AVG, P));
    CK(dalloc_buf<float>(c, PPO_BUF_EXP_AVG_SQ, P));
    CK(dalloc_buf<float>(c, PPO_BUF_ENV_STATE, N * c->O));
    CK(dalloc_buf<int32_t>(c, PPO_BUF_EP_LEN, N));
    CK(dalloc_buf<float>(c, PPO_BUF_EP_REW, N));
    CK(dalloc_buf<int32_t>(c, PPO_BUF_RESET_COUNT, N));
    CK(dalloc_buf<int32_t>(c, PPO_BUF_PERM, (size_t)cfg->update_epochs * B));
    CK(dalloc_buf<int32_t>(c, PPO_BUF_FIN_LEN, TN));
    CK(dalloc_buf<float>(c, PPO_BUF_FIN_REW, TN));
    CK(dalloc(c, &c->error_flag, 1));
    CK(dalloc(c, &c->wr_dev, 8));   // [0..2] running maxima, [4..6] what the host mirror holds
    CK(hipHostMalloc(reinterpret_cast<void**>(&c->wr_host), 4 * sizeof(uint32_t), hipHostMallocMapped));
    CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->wr_host_dev), c->wr_host, 0));
    CK(hipStreamCreateWithFlags(&c->wr_stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; i++) c->wr_host[i] = 0u;
    c->use_mfma = A <= 4;   // the matrix-core update kernel folds heads of up to 4 logits; wider policies (2 x 64 nets) run the vector kernel
    // ppo_config.kernel_flags (include/ppo_hip.h): the only switch between kernels; nothing is read from the environment
    if (cfg->kernel_flags & PPO_KERNEL_UPDATE_VECTOR) c->use_mfma = false;
    c->update_single_wave = (cfg->kernel_flags & PPO_KERNEL_UPDATE_ONE_WAVE) != 0;
    c->rollout_vector = (cfg->kernel_flags & PPO_KERNEL_ROLLOUT_VECTOR) != 0;
    c->max_blocks_per_net = 512;
    const int Pmax = std::max(c->L.net_size[0], c->L.net_size[1]);
    CK(dalloc(c, &c->slab, (size_t)2 * c->max_blocks_per_net * Pmax));
    CK(dalloc(c, &c->stat_slab, (size_t)2 * c->max_blocks_per_net * 8));
    CK(dalloc(c, &c->loss_sums, 8));
    {   // [gstats | adv_stats] contiguous: a sharded update sends both in ONE all-reduce
        double* blk = nullptr;
        CK(dalloc(c, &blk, (size_t)PPO_GSTAT_DOUBLES + ((size_t)c->steps_per_update + 1) * PPO_ADV_PARTS * (sizeof(AdvStat) / sizeof(double))));
        c->gstats = blk;
        c->adv_stats = reinterpret_cast<AdvStat*>(blk + PPO_GSTAT_DOUBLES);
        CK(dalloc(c, &c->adv_norm, (size_t)c->steps_per_update + 1));
    }
    CK(dalloc(c, &c->adam_coefs, (size_t)c->steps_per_update + 1));
    CK(hipHostMalloc(reinterpret_cast<void**>(&c->adam_coefs_h), 2 * ((size_t)c->steps_per_update + 1) * sizeof(AdamCoef)));
    CK(hipHostMalloc(reinterpret_cast<void**>(&c->snap), 2 * sizeof(StatsSnap)));
    CK(hipEventCreateWithFlags(&c->snap_ev[0], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&c->snap_ev[1], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&c->coef_copied[0], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&c->coef_copied[1], hipEventDisableTiming));
    CK(dalloc(c, &c->step_stats, (size_t)c->steps_per_update + 1));
    CK(dalloc(c, &c->clipfrac_accum, 2));
    CK(dalloc(c, &c->norm2, 4 * GEN_MAX_LAYERS * GEN_NORM_PARTS));
    CK(dalloc(c, &c->fused_partial, (size_t)fused_opt_blocks(c->L) * 12));
    CK(dalloc(c, &c->ev_sums, PPO_EV_BLOCKS * 4));
    if (c->use_mfma && !generic) {
        CK(dalloc(c, &c->rec_critic, (size_t)B * 16));   // both nets' records of a sample side by side: one 64-byte sector (kernels_update_mfma.hip: pack_records_kernel)
        c->rec_actor = c->rec_critic + 8;
    }
    CK(dalloc(c, &c->row_counts, (size_t)c->T));
    CK(dalloc(c, &c->group_bits, (size_t)c->T * ((N + 63) / 64)));
    CK(dalloc(c, &c->ring, 1));
    CK(dalloc(c, &c->scratch_obs, N * c->O));
    CK(dalloc(c, &c->stamps, 24));
    if (c->gen) {
        GenericCtx& g = *c->gen;
        const GenLayout& GL = g.L;
        g.rows_max = (std::max<int64_t>(c->MB, c->N) + 127) / 128 * 128;
        const size_t R = (size_t)g.rows_max;
        g.bf16 = cfg->compute_dtype == PPO_DTYPE_BF16;
        if (g.bf16) {
            // every bf16 buffer: 128 extra rows (the unguarded staging prefetches up to three chunks past the last one it uses), pitches padded
            // to 128 columns, zero-filled once -- nothing ever writes the padding
            g.ld_in0 = (GL.obs + 127) / 128 * 128;
            g.ld_h = (GL.hidden + 127) / 128 * 128;
            const size_t RB = R + 128;
            CK(dalloc(c, &g.xin_bf, RB * g.ld_in0));
            if (gen_rows_packable(GL)) CK(dalloc(c, &g.row_rec, (size_t)B * 8));
            CK(dalloc(c, &g.obs_bf, ((size_t)B + 128) * g.ld_in0));   // the rollout's observations as bf16, once per update (gen_fwd_bwd): 201 MB at 2048 x 128 x 384
            for (int net = 0; net < 2; net++)
                for (int l = 0; l < GL.n_hidden; l++) CK(dalloc(c, &g.acts_bf[net][l], RB * g.ld_h));
            for (int i = 0; i < 2; i++) {
                CK(dalloc(c, &g.tmp_bf[i], RB * g.ld_h));
                CK(dalloc(c, &g.dz_bf[0][i], RB * g.ld_h));
                CK(dalloc(c, &g.dz_bf[1][i], RB * g.ld_h));
                CK(dalloc(c, &g.dout_bf[i], RB * 128));
            }
            g.cs_layer_stride = (int64_t)(R / 64 + 2) * g.ld_h;   // one row of column sums per 64-row tile at most (kernels_generic_bwd.hip: a row range is >= one tile)
            CK(dalloc(c, &g.cs_part[0], (size_t)GL.n_layers * g.cs_layer_stride));
            CK(dalloc(c, &g.cs_part[1], (size_t)GL.n_layers * g.cs_layer_stride));
            CK(dalloc(c, &g.head_db_part, (size_t)GEN_LOSS_BLOCKS * (GL.act + 1)));
            {   // one pair of doubles per slab-sum workgroup (256 gradient elements: kernels_generic.hip SLAB_EPB) of the largest layer, per layer and net
                int64_t most = 0;
                for (int net = 0; net < 2; net++)
                    for (int l = 0; l < GL.n_layers; l++) most = std::max<int64_t>(most, (int64_t)GL.out_dim[net][l] * GL.in_dim[l] + GL.out_dim[net][l]);
                g.sq_cap = (int)((most + 255) / 256);
                CK(dalloc(c, &g.sq_part, (size_t)2 * GL.n_layers * g.sq_cap * 2));
            }
        } else {
            for (int net = 0; net < 2; net++)
                for (int l = 0; l < GL.n_hidden; l++) CK(dalloc(c, &g.acts[net][l], R * GL.hidden));
            for (int i = 0; i < 2; i++) CK(dalloc(c, &g.dz[i], R * GL.hidden));
            CK(dalloc(c, &g.xin, R * GL.obs));
            CK(dalloc(c, &g.dlogits, R * GL.act));
            CK(dalloc(c, &g.dval, R));
        }
        CK(dalloc(c, &g.logits, R * GL.act));
        CK(dalloc(c, &g.val, R));
        for (int i = 0; i < 4; i++) CK(dalloc(c, &g.row_f[i], R));
        CK(dalloc(c, &g.row_f[4], R + 2));
        CK(dalloc(c, &g.row_act, R * GL.n_heads));
        CK(dalloc(c, &g.row_mask, R * GL.act));

        CK(dalloc(c, &g.loss_part, (size_t)GEN_LOSS_BLOCKS * 8));
        {
            int64_t mx = 0;
            for (int l = 0; l < GL.n_layers; l++) mx = std::max<int64_t>(mx, (int64_t)std::max(GL.out_dim[0][l], GL.out_dim[1][l]) * (GL.in_dim[l] + 1));
            g.wslab_stride = (mx + 3) & ~3ll;
            // bf16 storage: a block of slabs per layer (a net's slab sums are ONE launch behind its backward pass: 10 launches less per minibatch step);
            // 5 layers x 129 slabs x 386 KB x 2 nets = 0.5 GB of 288
            g.wslab_layer_stride = (int64_t)(GEN_SPLIT_MFMA + 1) * g.wslab_stride;
            CK(dalloc(c, &g.wslab, (size_t)(g.bf16 ? GL.n_layers : 1) * g.wslab_layer_stride));
            if (g.bf16) {
                CK(dalloc(c, &g.wslab1, (size_t)GL.n_layers * g.wslab_layer_stride));
                CK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
                CK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                CK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
                CK(hipEventCreateWithFlags(&c->ev_gather, hipEventDisableTiming));
            }
            CK(dalloc(c, &g.db_part, (size_t)GEN_DB_CHUNKS * std::max(GL.hidden, GL.act)));
        }
        {   // bf16 planes of every layer's weights, padded to tile multiples (kernels_gemm.hip: PlaneStage)
            int64_t off = 0;
            for (int net = 0; net < 2; net++)
                for (int l = 0; l < GL.n_layers; l++) {
                    g.wp_npad[net][l] = (GL.out_dim[net][l] + 127) / 128 * 128;
                    g.wp_kpad[l] = (GL.in_dim[l] + 127) / 128 * 128;   // the tiled kernel's chunking; the fused kernels walk it in passes of 8 k steps
                    g.wp_off[net][l] = off;
                    off += 3ll * g.wp_npad[net][l] * g.wp_kpad[l];
                }
            CK(dalloc(c, &g.wplanes, (size_t)off + 128 * 256));   // + slack for the staging's prefetch past the last chunk
            if (cfg->compute_dtype == PPO_DTYPE_BF16) CK(dalloc(c, &g.wfrags, (size_t)off + 128 * 256));
            g.planes_dirty = true;
        }
        CK(dalloc(c, &g.act64, N * GL.n_heads));
        CK(dalloc(c, &g.step_lp, N));
        CK(dalloc(c, &g.step_en, N));
        CK(dalloc(c, &c->cur_mask, N * GL.act));
        // the zero-fills above ran on the null stream, which a non-blocking stream does not wait for: drain them before the first write
        CK(hipDeviceSynchronize());
        CK(gen_fill(g.row_f[4] + 2, (int64_t)R, 1.0f, c->stream));   // the ones vector of the bias-gradient gemv
        // arithmetic of the layer products (kernels_gemm.hip): fp32 carried as three bf16 terms, or plain bf16 operands (configs[4])
        g.gemm_prec = cfg->compute_dtype == PPO_DTYPE_BF16 ? PPO_MM_BF16 : PPO_MM_F32X3;
    }
#undef CK
    // every env can reset at most once per step: steps per env over the whole run bounds the shared reset stream
    const int64_t steps_per_env = cfg->total_timesteps > 0 ? cfg->total_timesteps / std::max<int64_t>(c->cfg.global_num_envs, 1) : 0;
    ppo_status st = ensure_reset_table(c, std::max<int64_t>(steps_per_env + cfg->num_steps + 4, 4096));
    if (st != PPO_OK) { g_create_error = c->err; ppo_ctx_destroy(c); return st; }
    // the zero-fills of the allocations ran on the null stream, which the context's non-blocking stream does not wait for
    if (hipDeviceSynchronize() != hipSuccess) { fail(nullptr, PPO_ERR_HIP, "hipDeviceSynchronize failed at the end of ppo_ctx_create"); ppo_ctx_destroy(c); return PPO_ERR_HIP; }
    *out = c;
    return PPO_OK;
}

extern "C" ppo_status ppo_sync(ppo_ctx* c) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return comm_health(c);
}
extern "C" void* ppo_stream(ppo_ctx* c) { return c ? c->stream : nullptr; }
extern "C" ppo_status ppo_get_config(const ppo_ctx* c, ppo_config* out) {
    if (!c || !out) return PPO_ERR_INVALID;
    *out = c->cfg;
    return PPO_OK;
}
extern "C" ppo_status ppo_buffer(ppo_ctx* c, int32_t which, void** dev_ptr, size_t* bytes) {
    NEED(c, c != nullptr, "null ctx");
    NEED(c, which >= 0 && which < PPO_BUF_COUNT_, "unknown buffer id");
    if (dev_ptr) *dev_ptr = c->buf[which];
    if (bytes) *bytes = c->buf_bytes[which];
    return PPO_OK;
}
extern "C" ppo_status ppo_device_alloc(ppo_ctx* c, size_t bytes, void** dev_ptr) {
    NEED(c, c && dev_ptr, "null argument");
    DeviceGuard dev_guard(c);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMalloc(dev_ptr, std::max<size_t>(bytes, 16)));
    return PPO_OK;
}
extern "C" ppo_status ppo_device_free(ppo_ctx* c, void* dev_ptr) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(dev_ptr));
    return PPO_OK;
}
extern "C" ppo_status ppo_memcpy_h2d(ppo_ctx* c, void* dst_dev, const void* src_h, size_t bytes) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    HIPCHK(c, hipMemcpyAsync(dst_dev, src_h, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->gen) { c->gen->planes_dirty = true; c->gen_opt_fused_ok = false; }   // the copy may have landed in the parameters (re-split the weights before their next use) or in the gradient
    c->wrange_dirty = true;                    // ... and their fp16-range maxima are recomputed before the next matrix-core launch
    return PPO_OK;
}
extern "C" ppo_status ppo_memcpy_d2h(ppo_ctx* c, void* dst_h, const void* src_dev, size_t bytes) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    HIPCHK(c, hipMemcpyAsync(dst_h, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PPO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Agent
// ---------------------------------------------------------------------------------------------------------
extern "C" int64_t ppo_param_count(const ppo_ctx* c) { return c ? c->L.P : -1; }

extern "C" ppo_status ppo_param_shapes(const ppo_ctx* c, int64_t* shapes_h, int32_t* n_tensors) {
    if (!c) return PPO_ERR_INVALID;
    if (n_tensors) *n_tensors = c->L.n_tensors;
    if (shapes_h && c->gen) {
        const GenLayout& GL = c->gen->L;
        int k = 0;
        for (int net = 0; net < 2; net++)
            for (int l = 0; l < GL.n_layers; l++) {
                shapes_h[k++] = GL.out_dim[net][l]; shapes_h[k++] = GL.in_dim[l];
                shapes_h[k++] = GL.out_dim[net][l]; shapes_h[k++] = 1;
            }
        return PPO_OK;
    }
    if (shapes_h) {
        int k = 0;
        for (int net = 0; net < 2; net++) {
            const int out3 = net == 0 ? 1 : c->A;
            const int64_t dims[6][2] = { { PPO_HIDDEN, c->O }, { PPO_HIDDEN, 1 }, { PPO_HIDDEN, PPO_HIDDEN }, { PPO_HIDDEN, 1 }, { out3, PPO_HIDDEN }, { out3, 1 } };
            for (auto& d : dims) { shapes_h[k++] = d[0]; shapes_h[k++] = d[1]; }
        }
    }
    return PPO_OK;
}

extern "C" ppo_status ppo_params_set_h(ppo_ctx* c, const float* params_h, int64_t count) {
    NEED(c, c && params_h, "null argument");
    NEED(c, count == c->L.P, "parameter count mismatch");
    return ppo_memcpy_h2d(c, c->buf[PPO_BUF_PARAMS], params_h, (size_t)count * sizeof(float));
}
extern "C" ppo_status ppo_params_get_h(ppo_ctx* c, float* params_h, int64_t count) {
    NEED(c, c && params_h, "null argument");
    NEED(c, count == c->L.P, "parameter count mismatch");
    return ppo_memcpy_d2h(c, params_h, c->buf[PPO_BUF_PARAMS], (size_t)count * sizeof(float));
}
extern "C" ppo_status ppo_optimizer_set_h(ppo_ctx* c, const float* m_h, const float* v_h, int64_t count, int64_t step) {
    NEED(c, c && m_h && v_h, "null argument");
    NEED(c, count == c->L.P, "parameter count mismatch");
    ppo_status s = ppo_memcpy_h2d(c, c->buf[PPO_BUF_EXP_AVG], m_h, (size_t)count * sizeof(float));
    if (s != PPO_OK) return s;
    s = ppo_memcpy_h2d(c, c->buf[PPO_BUF_EXP_AVG_SQ], v_h, (size_t)count * sizeof(float));
    c->opt_step = step;
    return s;
}
extern "C" ppo_status ppo_optimizer_get_h(ppo_ctx* c, float* m_h, float* v_h, int64_t count, int64_t* step) {
    NEED(c, c != nullptr, "null ctx");
    NEED(c, count == c->L.P, "parameter count mismatch");
    ppo_status s = PPO_OK;
    if (m_h) s = ppo_memcpy_d2h(c, m_h, c->buf[PPO_BUF_EXP_AVG], (size_t)count * sizeof(float));
    if (s == PPO_OK && v_h) s = ppo_memcpy_d2h(c, v_h, c->buf[PPO_BUF_EXP_AVG_SQ], (size_t)count * sizeof(float));
    if (step) *step = c->opt_step;
    return s;
}

// Agent::ppoLayerInit (Agent.cpp:91-99): torch::nn::init::orthogonal_(W, gain) = gain * Q of a Gaussian matrix (QR with the
// sign of diag(R) folded in), constant_(bias, 0).  LibTorch draws the Gaussian from its global mt19937 and calls LAPACK;
// here: Box-Muller on Philox(seed; tensor, element) and Householder QR in binary64.  Distributionally equivalent, not bit-equal.
static void orthogonal_fill(float* W, int rows, int cols, double gain, int64_t seed, int tensor_id) {
    const bool transpose = rows < cols;
    const int m = transpose ? cols : rows, n = transpose ? rows : cols;  // m >= n
    std::vector<double> Aq((size_t)m * n);
    auto philox_host = [&](uint32_t c0, uint32_t c1, uint32_t out[4]) {
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)((uint64_t)seed >> 32), c2 = (uint32_t)tensor_id, c3 = 3u;
        for (int r = 0; r < 10; r++) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    };
    for (size_t i = 0; i < Aq.size(); i += 2) {
        uint32_t w[4];
        philox_host((uint32_t)i, (uint32_t)(i >> 32), w);
        const double u1 = ((double)w[0] + 1.0) / 4294967296.0, u2 = (double)w[1] / 4294967296.0;
        const double r = std::sqrt(-2.0 * std::log(u1));
        Aq[i] = r * std::cos(2.0 * M_PI * u2);
        if (i + 1 < Aq.size()) Aq[i + 1] = r * std::sin(2.0 * M_PI * u2);
    }
    // Householder QR of Aq (m x n, row-major); accumulate Q (m x n) explicitly.
    std::vector<double> R = Aq, Q((size_t)m * n, 0.0), v((size_t)m);
    std::vector<std::vector<double>> vs;
    for (int k = 0; k < n; k++) {
        double norm = 0.0;
        for (int i = k; i < m; i++) norm += R[(size_t)i * n + k] * R[(size_t)i * n + k];
        norm = std::sqrt(norm);
        std::fill(v.begin(), v.end(), 0.0);
        const double alpha = R[(size_t)k * n + k] >= 0 ? -norm : norm;
        for (int i = k; i < m; i++) v[i] = R[(size_t)i * n + k];
        v[k] -= alpha;
        double vn = 0.0;
        for (int i = k; i < m; i++) vn += v[i] * v[i];
        if (vn > 0) {
            for (int j = k; j < n; j++) {
                double dot = 0.0;
                for (int i = k; i < m; i++) dot += v[i] * R[(size_t)i * n + j];
                const double f = 2.0 * dot / vn;
                for (int i = k; i < m; i++) R[(size_t)i * n + j] -= f * v[i];
            }
        }
        vs.push_back(v);
    }
    for (int j = 0; j < n; j++) Q[(size_t)j * n + j] = 1.0;
    for (int k = n - 1; k >= 0; k--) {
        const std::vector<double>& vk = vs[k];
        double vn = 0.0;
        for (int i = k; i < m; i++) vn += vk[i] * vk[i];
        if (vn <= 0) continue;
        for (int j = 0; j < n; j++) {
            double dot = 0.0;
            for (int i = k; i < m; i++) dot += vk[i] * Q[(size_t)i * n + j];
            const double f = 2.0 * dot / vn;
            for (int i = k; i < m; i++) Q[(size_t)i * n + j] -= f * vk[i];
        }
    }
    for (int j = 0; j < n; j++) {
        const double sgn = R[(size_t)j * n + j] < 0 ? -1.0 : 1.0;  // q *= sign(diag(r))
        for (int i = 0; i < m; i++) Q[(size_t)i * n + j] *= sgn;
    }
    for (int r = 0; r < rows; r++)
        for (int cc = 0; cc < cols; cc++) W[(size_t)r * cols + cc] = (float)(gain * (transpose ? Q[(size_t)cc * n + r] : Q[(size_t)r * n + cc]));
}

extern "C" ppo_status ppo_params_init_orthogonal(ppo_ctx* c, int64_t seed) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    std::vector<float> p((size_t)c->L.P, 0.0f);
    if (c->gen) {   // Agent.cpp:25-59 generalised: sqrt(2) on hidden layers, 1.0 on the value head, 0.01 on the policy head
        const GenLayout& GL = c->gen->L;
        for (int net = 0; net < 2; net++)
            for (int l = 0; l < GL.n_layers; l++) {
                const double gain = l < GL.n_hidden ? std::sqrt(2.0) : (net == 0 ? 1.0 : 0.01);
                orthogonal_fill(p.data() + GL.w_off[net][l], GL.out_dim[net][l], GL.in_dim[l], gain, seed, net * GEN_MAX_LAYERS + l);
            }
        return ppo_params_set_h(c, p.data(), c->L.P);
    }
    for (int net = 0; net < 2; net++) {
        const int out3 = net == 0 ? 1 : c->A;
        orthogonal_fill(p.data() + c->L.w1[net], PPO_HIDDEN, c->O, std::sqrt(2.0), seed, net * 3 + 0);
        orthogonal_fill(p.data() + c->L.w2[net], PPO_HIDDEN, PPO_HIDDEN, std::sqrt(2.0), seed, net * 3 + 1);
        orthogonal_fill(p.data() + c->L.w3[net], out3, PPO_HIDDEN, net == 0 ? 1.0 : 0.01, seed, net * 3 + 2);  // Agent.cpp:28,37
    }
    return ppo_params_set_h(c, p.data(), c->L.P);
}

// ---- generic networks (generic.hpp): critic / actor over n rows, chunked to the workspace size ----
static ppo_status gen_values(ppo_ctx* c, const float* obs, int64_t n, float* value) {
    GenericCtx& g = *c->gen;
    if (gen_fused_forward_ok(g) && n <= GEN_FUSED_MAX_ROWS) {   // bf16 storage, small batch: the whole critic in one launch
        HIPCHK(c, gen_fused_forward(g, B_<float>(c, PPO_BUF_PARAMS), 0, obs, nullptr, 0, n, false, value, c->stream));
        return PPO_OK;
    }
    for (int64_t off = 0; off < n; off += g.rows_max) {
        const int64_t rows = std::min<int64_t>(g.rows_max, n - off);
        HIPCHK(c, gen_forward(g, B_<float>(c, PPO_BUF_PARAMS), 0, obs + off * g.L.obs, rows, nullptr, g.dz[0], g.dz[1], value + off, c->stream));
    }
    return PPO_OK;
}
static ppo_status gen_policy(ppo_ctx* c, const float* obs, const uint8_t* mask, const int64_t* forced, int64_t n, int64_t step_index, int64_t* action,
                             float* logprob, float* entropy) {
    GenericCtx& g = *c->gen;
    for (int64_t off = 0; off < n; off += g.rows_max) {
        const int64_t rows = std::min<int64_t>(g.rows_max, n - off);
        if (gen_fused_forward_ok(g) && rows <= GEN_FUSED_MAX_ROWS) HIPCHK(c, gen_fused_forward(g, B_<float>(c, PPO_BUF_PARAMS), 1, obs + off * g.L.obs, nullptr, 0, rows, false, g.logits, c->stream));
        else HIPCHK(c, gen_forward(g, B_<float>(c, PPO_BUF_PARAMS), 1, obs + off * g.L.obs, rows, nullptr, g.dz[0], g.dz[1], g.logits, c->stream));
        HIPCHK(c, gen_heads(g.L, c->cfg.dist_kind, g.logits, mask ? mask + off * g.L.act : nullptr, forced ? forced + off * g.L.n_heads : nullptr, rows,
                            c->cfg.seed, c->cfg.env_offset + off, step_index, action ? action + off * g.L.n_heads : nullptr,
                            logprob ? logprob + off : nullptr, entropy ? entropy + off : nullptr, c->stream));
    }
    return PPO_OK;
}

extern "C" ppo_status ppo_get_value(ppo_ctx* c, const float* obs, int64_t n, float* value) {
    NEED(c, c && obs && value, "null argument");
    DeviceGuard dev_guard(c);
    if (c->gen) return gen_values(c, obs, n, value);
    HIPCHK(c, launch_policy_act(B_<float>(c, PPO_BUF_PARAMS), c->L, c->cfg.dist_kind, obs, nullptr, nullptr, n, c->cfg.seed, c->cfg.env_offset, 0,
                                nullptr, nullptr, nullptr, value, true, c->stream));
    return PPO_OK;
}

static ppo_status refresh_weight_range(ppo_ctx* c);          // (the fp16 range of the matrix-core kernels' operands: defined with ppo_rollout below)
static inline void wr_snapshot(ppo_ctx* c);
static inline bool weights_fit_rollout16(const ppo_ctx* c);

extern "C" ppo_status ppo_policy_act(ppo_ctx* c, const float* obs, const uint8_t* mask, const int64_t* forced_action, int64_t n,
                                     int64_t step_index, int64_t* action, float* logprob, float* entropy, float* value) {
    NEED(c, c && obs, "null argument");
    DeviceGuard dev_guard(c);
    NEED(c, forced_action || action, "sampling needs an action output");
    if (c->gen) {
        ppo_status s = gen_policy(c, obs, mask, forced_action, n, step_index, action, logprob, entropy);
        if (s == PPO_OK && value) s = gen_values(c, obs, n, value);
        return s;
    }
    // The stand-alone policy runs the arithmetic of the kernel this context's ROLLOUT would run now (round 6): rollout16_kernel's two-term fp16 products
    // on the matrix cores by default, the vector ALU's fp32 multiply-adds under PPO_KERNEL_ROLLOUT_VECTOR or while the output layer's weights do not fit
    // (the same range snapshot, the same decision as ppo_rollout) -- so a caller comparing the two on the same observations sees the same bits.
    bool as16 = !c->rollout_vector && policy_act16_serves(c->L);
    if (as16) {
        const ppo_status rs = refresh_weight_range(c);
        if (rs != PPO_OK) return rs;
        wr_snapshot(c);
        if (!weights_fit_rollout16(c)) { as16 = false; c->vector_fallback_launches += 1; }
    }
    HIPCHK(c, launch_policy_act(B_<float>(c, PPO_BUF_PARAMS), c->L, c->cfg.dist_kind, obs, mask, forced_action, n, c->cfg.seed, c->cfg.env_offset,
                                step_index, action, logprob, entropy, value, false, c->stream, as16, c->error_flag));
    return PPO_OK;
}

extern "C" ppo_status ppo_categorical(int32_t dist_kind, const float* logits, const uint8_t* mask, const int64_t* value, int64_t n, int32_t A,
                                      float* m_logits, float* m_probs, float* log_prob, float* entropy, int64_t* mode, void* stream) {
    if (!logits || n < 0) return PPO_ERR_INVALID;
    return launch_categorical(dist_kind, logits, mask, value, n, A, m_logits, m_probs, log_prob, entropy, mode, (hipStream_t)stream) == hipSuccess ? PPO_OK : PPO_ERR_HIP;
}

extern "C" ppo_status ppo_categorical_sample(const float* m_probs, int64_t n, int32_t A, int64_t seed, int64_t row_offset, int64_t step_index,
                                             int32_t head, int64_t* sample, void* stream) {
    if (!m_probs || !sample || n < 0) return PPO_ERR_INVALID;
    return launch_categorical_sample(m_probs, n, A, seed, row_offset, step_index, head, sample, (hipStream_t)stream) == hipSuccess ? PPO_OK : PPO_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------
// Environments
// ---------------------------------------------------------------------------------------------------------
extern "C" ppo_status ppo_env_transition(int32_t env_kind, const float* state_in, const int64_t* action, int64_t n, float* next_state,
                                         float* reward, int32_t* terminated, void* stream) {
    if (!state_in || !action || !next_state || !reward || !terminated || n < 0) return PPO_ERR_INVALID;
    return launch_env_transition(env_kind, state_in, action, n, next_state, reward, terminated, (hipStream_t)stream) == hipSuccess ? PPO_OK : PPO_ERR_HIP;
}

extern "C" ppo_status ppo_env_reset(ppo_ctx* c) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    if (c->gen) {   // synthetic env: memoryless, the observation of global step `rollout_steps` (0 after creation)
        HIPCHK(c, hipMemsetAsync(c->buf[PPO_BUF_EP_LEN], 0, (size_t)c->N * sizeof(int32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(c->buf[PPO_BUF_EP_REW], 0, (size_t)c->N * sizeof(float), c->stream));
        HIPCHK(c, hipMemsetAsync(c->buf[PPO_BUF_NEXT_DONE], 0, (size_t)c->N * sizeof(int32_t), c->stream));
        HIPCHK(c, gen_synthetic_step(c->gen->L, c->N, c->cfg.seed, c->cfg.env_offset, c->rollout_steps, c->cfg.max_episode_steps, nullptr, nullptr,
                                     B_<float>(c, PPO_BUF_NEXT_OBS), c->cur_mask, nullptr, nullptr, nullptr, nullptr, c->stream));
        return PPO_OK;
    }
    HIPCHK(c, launch_env_reset(c->cfg.env_kind, c->N, c->cfg.seed, c->cfg.env_offset, B_<float>(c, PPO_BUF_ENV_STATE), B_<int32_t>(c, PPO_BUF_EP_LEN),
                               B_<float>(c, PPO_BUF_EP_REW), B_<int32_t>(c, PPO_BUF_RESET_COUNT), c->reset_table, c->reset_cap,
                               B_<float>(c, PPO_BUF_NEXT_OBS), B_<int32_t>(c, PPO_BUF_NEXT_DONE), c->error_flag, c->stream));
    return PPO_OK;
}

extern "C" ppo_status ppo_env_step(ppo_ctx* c, const int64_t* action, float* obs, float* reward, int32_t* done) {
    NEED(c, c && action && obs && reward && done, "null argument");
    DeviceGuard dev_guard(c);
    if (c->gen) {   // synthetic env: one step at the context's global step counter (the action does not influence it)
        HIPCHK(c, gen_synthetic_step(c->gen->L, c->N, c->cfg.seed, c->cfg.env_offset, c->rollout_steps, c->cfg.max_episode_steps,
                                     B_<int32_t>(c, PPO_BUF_EP_LEN), B_<float>(c, PPO_BUF_EP_REW), obs, c->cur_mask, reward, done, nullptr, nullptr, c->stream));
        c->rollout_steps += 1;
        return PPO_OK;
    }
    HIPCHK(c, launch_env_step(c->cfg.env_kind, c->N, c->H, c->cfg.max_episode_steps, c->cfg.seed, c->cfg.env_offset, B_<float>(c, PPO_BUF_ENV_STATE),
                              B_<int32_t>(c, PPO_BUF_EP_LEN), B_<float>(c, PPO_BUF_EP_REW), B_<int32_t>(c, PPO_BUF_RESET_COUNT), c->reset_table,
                              c->reset_cap, action, obs, reward, done, c->error_flag, c->stream));
    return PPO_OK;
}

extern "C" ppo_status ppo_env_set_state_h(ppo_ctx* c, const float* state_h, const int32_t* ep_len_h, const float* ep_rew_h, const int32_t* reset_count_h) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    if (state_h) {
        ppo_status s = ppo_memcpy_h2d(c, c->scratch_obs, state_h, (size_t)c->N * c->O * sizeof(float));
        if (s != PPO_OK) return s;
        HIPCHK(c, launch_aos_to_soa(c->scratch_obs, B_<float>(c, PPO_BUF_ENV_STATE), c->N, c->O, true, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->buf[PPO_BUF_NEXT_OBS], c->scratch_obs, (size_t)c->N * c->O * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    ppo_status s = PPO_OK;
    if (ep_len_h) s = ppo_memcpy_h2d(c, c->buf[PPO_BUF_EP_LEN], ep_len_h, (size_t)c->N * sizeof(int32_t));
    if (s == PPO_OK && ep_rew_h) s = ppo_memcpy_h2d(c, c->buf[PPO_BUF_EP_REW], ep_rew_h, (size_t)c->N * sizeof(float));
    if (s == PPO_OK && reset_count_h) s = ppo_memcpy_h2d(c, c->buf[PPO_BUF_RESET_COUNT], reset_count_h, (size_t)c->N * sizeof(int32_t));
    if (s == PPO_OK) HIPCHK(c, hipStreamSynchronize(c->stream));
    return s;
}

extern "C" ppo_status ppo_env_get_state_h(ppo_ctx* c, float* state_h, int32_t* ep_len_h, float* ep_rew_h, int32_t* reset_count_h) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    ppo_status s = PPO_OK;
    if (state_h) {
        HIPCHK(c, launch_aos_to_soa(c->scratch_obs, B_<float>(c, PPO_BUF_ENV_STATE), c->N, c->O, false, c->stream));
        s = ppo_memcpy_d2h(c, state_h, c->scratch_obs, (size_t)c->N * c->O * sizeof(float));
    }
    if (s == PPO_OK && ep_len_h) s = ppo_memcpy_d2h(c, ep_len_h, c->buf[PPO_BUF_EP_LEN], (size_t)c->N * sizeof(int32_t));
    if (s == PPO_OK && ep_rew_h) s = ppo_memcpy_d2h(c, ep_rew_h, c->buf[PPO_BUF_EP_REW], (size_t)c->N * sizeof(float));
    if (s == PPO_OK && reset_count_h) s = ppo_memcpy_d2h(c, reset_count_h, c->buf[PPO_BUF_RESET_COUNT], (size_t)c->N * sizeof(int32_t));
    return s;
}

// ---------------------------------------------------------------------------------------------------------
// Rollout and advantages
// ---------------------------------------------------------------------------------------------------------
static ppo_status gen_rollout(ppo_ctx* c, const int64_t* forced);
static ppo_status consume_finished_episodes(ppo_ctx* c) {
    if (!c->fin_pending) return PPO_OK;
    // the rollout that left these episodes has already advanced rollout_steps by T
    HIPCHK(c, launch_episode_ring_update(B_<int32_t>(c, PPO_BUF_FIN_LEN), B_<float>(c, PPO_BUF_FIN_REW), c->T, c->N, c->row_counts, c->group_bits, c->ring,
                                         c->rollout_steps - c->T, c->cfg.global_num_envs, c->cfg.env_offset, c->stream));
    c->fin_pending = false;
    return PPO_OK;
}

// The matrix-core kernels of the reference's two shapes carry some operands as fp16 (kernels_rollout.hip: 2^8 W3; kernels_update_mfma.hip: c W2 and the
// products through its columns); the reference has no such limits.  A drop-in user never sees them: weight_range_kernel takes the maxima of |parameter| per
// class once per update on a stream of its own (sweep_weight_range below; ppo_internal.hpp has the cost bisect), the host reads the pinned mirror -- no
// synchronisation; it lags the device by the launches in flight, during which AdamW moves a weight by about lr per step, hence thresholds at HALF the
// kernels' limits -- and a launch whose weights are out of range takes the vector kernel (plain fp32, same function; counted in
// ppo_profile.vector_fallback_launches).  After the host wrote parameters the maxima are recomputed from scratch (one tiny launch + a synchronisation).
constexpr float WR_LIMIT_W3 = 128.0f;    // rollout16_kernel: |W3| < 255
constexpr float WR_LIMIT_W2 = 4.0f;      // update kernels: a column of c W2 with absolute sum ~2^10 overflows the fp16 terms of dz1: 64 x 4 x 2.885 = 739
constexpr float WR_LIMIT_REST = 8192.0f; // c W1, c b1, c b2: < 65 504 / 2.885
static inline void wr_snapshot(ppo_ctx* c) {
    for (int i = 0; i < 3; i++) {
        const uint32_t bits = __atomic_load_n(c->wr_host + i, __ATOMIC_RELAXED);
        std::memcpy(&c->wr_cache[i], &bits, 4);   // NaN (a diverged run) compares false below: not in range
    }
}
static ppo_status refresh_weight_range(ppo_ctx* c) {
    if (!c->wrange_dirty || c->gen) return PPO_OK;
    HIPCHK(c, launch_weight_range(B_<float>(c, PPO_BUF_PARAMS), c->L, c->wr_dev, c->wr_host_dev, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->wrange_dirty = false;
    wr_snapshot(c);
    return PPO_OK;
}
// The maxima again, once per update (and after a stand-alone optimizer step): one tiny launch on a stream of its own with NO dependency on the update's
// stream -- it reads whatever the parameters are when it runs (a 32-bit load each; AdamW may be writing them), which is the lag the thresholds' margin
// already covers, and costs the update's stream nothing (as the tail of pack_records_kernel the same sweep cost 7 us per update; inside the AdamW
// kernels, 40 times that).
static ppo_status sweep_weight_range(ppo_ctx* c, hipStream_t s) {
    if (c->gen || c->wrange_dirty) return PPO_OK;   // dirty: the next matrix-core launch recomputes and waits (refresh_weight_range)
    HIPCHK(c, launch_weight_range(B_<float>(c, PPO_BUF_PARAMS), c->L, c->wr_dev, c->wr_host_dev, s));
    return PPO_OK;
}
static inline bool weights_fit_rollout16(const ppo_ctx* c) { return c->wr_cache[PPO_WR_W3] < WR_LIMIT_W3; }
static inline bool weights_fit_update_mfma(const ppo_ctx* c) {
    return c->wr_cache[PPO_WR_W3] < WR_LIMIT_REST && c->wr_cache[PPO_WR_W2] < WR_LIMIT_W2 && c->wr_cache[PPO_WR_REST] < WR_LIMIT_REST;
}
static inline OptGuard opt_guard(const ppo_ctx* c) {
    OptGuard g;
    g.error_flag = c->error_flag;
    return g;
}

extern "C" ppo_status ppo_rollout(ppo_ctx* c, const int64_t* forced_actions) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    ppo_status s = consume_finished_episodes(c);
    if (s != PPO_OK) return s;
    if (c->gen) return gen_rollout(c, forced_actions);
    // worst case one reset per env per step
    s = ensure_reset_table(c, c->rollout_steps + c->T + 4);
    if (s != PPO_OK) return s;
    RolloutArgs a{};
    a.params = B_<float>(c, PPO_BUF_PARAMS);
    a.L = c->L;
    a.dist_kind = c->cfg.dist_kind;
    a.env_kind = c->cfg.env_kind;
    a.N = c->N; a.T = c->T;
    a.max_episode_steps = c->cfg.max_episode_steps;
    a.seed = c->cfg.seed; a.env_offset = c->cfg.env_offset; a.step_base = c->rollout_steps;
    a.env_state = B_<float>(c, PPO_BUF_ENV_STATE);
    a.ep_len = B_<int32_t>(c, PPO_BUF_EP_LEN);
    a.ep_rew = B_<float>(c, PPO_BUF_EP_REW);
    a.reset_count = B_<int32_t>(c, PPO_BUF_RESET_COUNT);
    a.reset_table = c->reset_table; a.reset_cap = c->reset_cap; a.error_flag = c->error_flag;
    s = refresh_weight_range(c);
    if (s != PPO_OK) return s;
    a.vector_kernel = c->rollout_vector ? 1 : 0;
    wr_snapshot(c);
    if (!a.vector_kernel && !weights_fit_rollout16(c)) { a.vector_kernel = 1; c->vector_fallback_launches += 1; }
    a.obs = B_<float>(c, PPO_BUF_OBS);
    a.actions = B_<int32_t>(c, PPO_BUF_ACTIONS);
    a.logprobs = B_<float>(c, PPO_BUF_LOGPROBS);
    a.rewards = B_<float>(c, PPO_BUF_REWARDS);
    a.dones = B_<float>(c, PPO_BUF_DONES);
    a.values = B_<float>(c, PPO_BUF_VALUES);
    a.masks = c->cfg.dist_kind == PPO_DIST_MASKED ? B_<uint8_t>(c, PPO_BUF_MASKS) : nullptr;
    a.fin_len = B_<int32_t>(c, PPO_BUF_FIN_LEN);
    a.fin_rew = B_<float>(c, PPO_BUF_FIN_REW);
    a.next_obs = B_<float>(c, PPO_BUF_NEXT_OBS);
    a.next_done = B_<int32_t>(c, PPO_BUF_NEXT_DONE);
    a.next_value = B_<float>(c, PPO_BUF_NEXT_VALUE);
    a.forced_actions = forced_actions;
    {
        ProfScope ps(c, PROF_ROLLOUT);
        HIPCHK(c, launch_rollout(a, c->stream));
    }
    c->rollout_steps += c->T;
    c->global_step += (int64_t)c->T * c->cfg.global_num_envs;  // global_step += num_envs per step (:526)
    c->fin_pending = true;
    return PPO_OK;
}

// Stateless Linear-layer product on the matrix cores (kernels_gemm.hip).
extern "C" ppo_status ppo_matmul(int32_t trans_a, int32_t trans_b, int64_t M, int64_t N, int64_t K, const float* a, int64_t lda, const float* b, int64_t ldb,
                                 float* c, int64_t ldc, int32_t epilogue, const float* aux, int64_t ld_aux, int32_t precision, void* stream) {
    if (!a || !b || !c || M < 0 || N < 0 || K < 0 || epilogue < PPO_MM_EPI_NONE || epilogue > PPO_MM_EPI_DTANH) return PPO_ERR_INVALID;
    if (epilogue != PPO_MM_EPI_NONE && !aux) return PPO_ERR_INVALID;
    if (precision != PPO_MM_F32X3 && precision != PPO_MM_BF16) return PPO_ERR_INVALID;
    return launch_matmul(trans_a != 0, trans_b != 0, M, N, K, a, lda, b, ldb, c, ldc, epilogue, aux, ld_aux, precision, 1, 0, nullptr, 0, nullptr, 0, 0, (hipStream_t)stream) == hipSuccess
               ? PPO_OK : PPO_ERR_HIP;
}
extern "C" ppo_status ppo_gae(const float* rewards, const float* values, const float* dones, const float* next_value, const int32_t* next_done,
                              int64_t T, int64_t N, float gamma, float gae_lambda, float* advantages, float* returns, void* stream) {
    if (!rewards || !values || !dones || !next_value || !next_done || !advantages || !returns || T < 0 || N < 0) return PPO_ERR_INVALID;
    return launch_gae(rewards, values, dones, next_value, next_done, T, N, gamma, gae_lambda, advantages, returns, nullptr, (hipStream_t)stream) == hipSuccess ? PPO_OK : PPO_ERR_HIP;
}
extern "C" ppo_status ppo_gae_fast(const float* rewards, const float* values, const float* dones, const float* next_value, const int32_t* next_done,
                                   int64_t T, int64_t N, float gamma, float gae_lambda, float* advantages, float* returns, void* stream) {
    if (!rewards || !values || !dones || !next_value || !next_done || !advantages || !returns || T < 0 || N < 0) return PPO_ERR_INVALID;
    return launch_gae_fast(rewards, values, dones, next_value, next_done, T, N, gamma, gae_lambda, advantages, returns, (hipStream_t)stream) == hipSuccess ? PPO_OK : PPO_ERR_HIP;
}
extern "C" ppo_status ppo_nstep_returns(const float* rewards, const float* values, const float* dones, const float* next_value,
                                        const int32_t* next_done, int64_t T, int64_t N, float gamma, float* advantages, float* returns, void* stream) {
    if (!rewards || !values || !dones || !next_value || !next_done || !advantages || !returns || T < 0 || N < 0) return PPO_ERR_INVALID;
    return launch_nstep(rewards, values, dones, next_value, next_done, T, N, gamma, advantages, returns, nullptr, (hipStream_t)stream) == hipSuccess ? PPO_OK : PPO_ERR_HIP;
}

// K4 on the context's buffers: GAE (PPO_Discrete.cpp:283-306) or n-step returns (:309-329) by cfg.use_gae.
static ppo_status run_scan(ppo_ctx* c) {
    ProfScope ps(c, PROF_GAE);
    if (c->cfg.use_gae)
        HIPCHK(c, launch_gae(B_<float>(c, PPO_BUF_REWARDS), B_<float>(c, PPO_BUF_VALUES), B_<float>(c, PPO_BUF_DONES), B_<float>(c, PPO_BUF_NEXT_VALUE),
                             B_<int32_t>(c, PPO_BUF_NEXT_DONE), c->T, c->N, c->cfg.gamma, c->cfg.gae_lambda, B_<float>(c, PPO_BUF_ADVANTAGES),
                             B_<float>(c, PPO_BUF_RETURNS), c->error_flag, c->stream));
    else
        HIPCHK(c, launch_nstep(B_<float>(c, PPO_BUF_REWARDS), B_<float>(c, PPO_BUF_VALUES), B_<float>(c, PPO_BUF_DONES), B_<float>(c, PPO_BUF_NEXT_VALUE),
                               B_<int32_t>(c, PPO_BUF_NEXT_DONE), c->T, c->N, c->cfg.gamma, B_<float>(c, PPO_BUF_ADVANTAGES),
                               B_<float>(c, PPO_BUF_RETURNS), c->error_flag, c->stream));
    return PPO_OK;
}

extern "C" ppo_status ppo_calc_advantage(ppo_ctx* c) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    // bootstrap value if not done: next_value = Critic(next_obs) (PPO_Discrete.cpp:280)
    const ppo_status s = ppo_get_value(c, B_<float>(c, PPO_BUF_NEXT_OBS), c->N, B_<float>(c, PPO_BUF_NEXT_VALUE));
    if (s != PPO_OK) return s;
    return run_scan(c);
}

// ---------------------------------------------------------------------------------------------------------
// Update
// ---------------------------------------------------------------------------------------------------------
extern "C" ppo_status ppo_generate_permutations(ppo_ctx* c) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    HIPCHK(c, launch_permutations(B_<int32_t>(c, PPO_BUF_PERM), c->B, c->cfg.update_epochs, c->cfg.seed, c->updates, c->rank, c->stream));
    return PPO_OK;
}

// AdamW scalars exactly as LibTorch forms them on the host (optim/adamw.cpp): doubles narrowed to float at use.
static AdamCoef adam_coef(double lr, int64_t t) {
    const double beta1 = 0.9, beta2 = 0.999, wd = 1e-2;
    const double bc1 = 1.0 - std::pow(beta1, (double)t), bc2 = 1.0 - std::pow(beta2, (double)t);
    AdamCoef k;
    k.decay = (float)(1.0 - lr * wd);
    k.neg_step = (float)(-(lr / bc1));
    k.sqrt_bc2 = (float)std::sqrt(bc2);
    k.pad = 0.0f;
    return k;
}

static ppo_status allreduce_sum(ppo_ctx* c, void* buf, size_t count, bool f64) {
    if (c->world <= 1 && !c->force_collectives) return PPO_OK;
    // bracketed like the dominant kernel: every call in mode 1, otherwise the one that follows a bracketed update launch
    ProfScope ps(c, PROF_ALLREDUCE, c->prof_every <= 1 || c->prof_last_sampled);
    c->prof_last_sampled = false;
    if (c->xchg) {
        ExchangeComm& x = *c->xchg;
        const size_t bytes = count * (f64 ? 8 : 4);
        NEED(c, bytes <= x.slot_bytes, "all-reduce payload exceeds the exchange slot");
        XchgPtrs pp{};
        for (int r = 0; r < c->world; r++) pp.p[r] = x.peer[r];
        x.seq += 1;
        HIPCHK(c, launch_exchange_allreduce(buf, count, f64, pp, c->rank, c->world, x.slot_bytes, x.seq, x.timeout_flag, c->stream));
        return PPO_OK;
    }
    if (c->lgroup) {
        LocalGroup* g = c->lgroup.get();
        HIPCHK(c, hipEventRecord(c->lg_ready, c->stream));
        uint64_t my_gen;
        hipError_t he = hipSuccess;
        bool failed = false;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            if (g->failed) return fail(c, PPO_ERR_COMM, "in-process group has failed");
            g->bufs[c->rank] = buf;
            g->ready[c->rank] = c->lg_ready;
            my_gen = g->generation;
            if (++g->arrived == g->n) {
                // last arriver: one kernel on its stream reads every rank's buffer.  Whatever happens here, the generation advances and everybody is
                // woken: a failure is reported to all members instead of leaving them in cv.wait
                for (int r = 0; r < g->n && he == hipSuccess; r++) he = hipStreamWaitEvent(c->stream, g->ready[r], 0);
                PtrPack pk{};
                for (int r = 0; r < g->n; r++) pk.p[r] = g->bufs[r];
                if (he == hipSuccess) he = launch_local_allreduce(pk, g->n, count, f64, c->stream);
                if (he == hipSuccess) he = hipEventRecord(g->done[my_gen & 1], c->stream);
                if (he != hipSuccess) g->failed = true;
                g->arrived = 0;
                g->generation++;
                g->cv.notify_all();
            } else {
                g->cv.wait(lk, [&] { return g->generation != my_gen; });
            }
            failed = g->failed;
        }
        if (he != hipSuccess) return fail(c, PPO_ERR_HIP, "in-process all-reduce failed: %s", hipGetErrorString(he));
        if (failed) return fail(c, PPO_ERR_COMM, "in-process all-reduce failed on another member of the group");
        HIPCHK(c, hipStreamWaitEvent(c->stream, g->done[my_gen & 1], 0));
        return PPO_OK;
    }
    const int rc = rccl::AllReduce(buf, buf, count, f64 ? rccl::kFloat64 : rccl::kFloat32, rccl::kSum, c->comm, c->stream);
    if (rc != 0) return fail(c, PPO_ERR_COMM, "ncclAllReduce failed: %s", rccl::GetErrorString ? rccl::GetErrorString(rc) : "?");
    return PPO_OK;
}

// ---- generic networks: one minibatch step = gather, two forward passes that keep their activations, loss, two backward passes ----
static ppo_status gen_fwd_bwd(ppo_ctx* c, const int32_t* idx, int64_t M, int slot) {
    GenericCtx& g = *c->gen;
    const GenLayout& GL = g.L;
    NEED(c, M >= 1 && M <= g.rows_max, "minibatch larger than the workspace");
    const double global_M = (double)M * c->world;
    c->last_global_M = global_M;
    float* params = B_<float>(c, PPO_BUF_PARAMS);
    float* grads = B_<float>(c, PPO_BUF_GRADS);
    {
        ProfScope ps(c, PROF_FWD_BWD);
        // Both fused passes available (bf16 storage, widths the kernels are built for): the two nets share every launch -- forward, each layer's backward, the
        // slab sums -- on ONE stream: the second net's workgroups take the CUs the first net's leave (forward) or run beside them on the other half of the
        // chip (backward), with no fork / join events between the streams (four per step, several us each).  Otherwise: the critic's passes on a second stream.
        const bool classic = (c->cfg.kernel_flags & PPO_KERNEL_GENERIC_CLASSIC) != 0;   // the step's first form (ppo_hip.h): A/B runs and tests
        const bool paired = !classic && g.bf16 && gen_fused_forward_ok(g) && gen_fused_backward_ok(g) && M <= GEN_FUSED_MAX_ROWS;
        const bool paired_bwd = paired;
        const bool two = !paired && g.bf16 && c->stream2 != nullptr;   // a kernel of one net fills the CUs the other net's kernel is draining
        const bool two_bwd = !paired_bwd && g.bf16 && c->stream2 != nullptr;
        auto gather = [&](const int32_t* rows, int64_t n, hipStream_t st) {
            return gen_gather(GL, B_<float>(c, PPO_BUF_OBS), B_<int32_t>(c, PPO_BUF_ACTIONS), B_<uint8_t>(c, PPO_BUF_MASKS), B_<float>(c, PPO_BUF_LOGPROBS),
                              B_<float>(c, PPO_BUF_ADVANTAGES), B_<float>(c, PPO_BUF_RETURNS), B_<float>(c, PPO_BUF_VALUES), rows, n, g, st);
        };
        // the fused kernels read the minibatch's rows in place (generic.hpp: GenericCtx::obs_bf, rows_idx); the observations are rounded once per update --
        // a stand-alone step (whose caller may have rewritten the buffers) rounds them every time
        const bool in_place = !classic && g.bf16 && g.obs_bf && gen_fused_forward_ok(g) && gen_fused_backward_ok(g) && M <= GEN_FUSED_MAX_ROWS;
        g.rows_idx = nullptr;
        g.rows_rec = nullptr;
        if (in_place) {
            g.rows_src = GenRowSrc{ B_<int32_t>(c, PPO_BUF_ACTIONS), c->cfg.dist_kind == PPO_DIST_MASKED ? B_<uint8_t>(c, PPO_BUF_MASKS) : nullptr, B_<float>(c, PPO_BUF_LOGPROBS),
                                    B_<float>(c, PPO_BUF_ADVANTAGES), B_<float>(c, PPO_BUF_RETURNS), B_<float>(c, PPO_BUF_VALUES) };
            if (!c->gen_obs_bf_valid) {
                // inside ppo_update: the whole batch once, valid for all of the update's steps.  A stand-alone step (its caller may have rewritten any buffer
                // since the last one): only the step's own M rows, in place -- not all B of them for a step that reads a few
                const bool whole = c->wr_in_update;
                HIPCHK(c, launch_to_bf16_pad(B_<float>(c, PPO_BUF_OBS), whole ? c->B : M, GL.obs, g.obs_bf, g.ld_in0, c->stream, whole ? nullptr : idx));
                if (g.row_rec) HIPCHK(c, gen_pack_rows(GL, g.rows_src, whole ? c->B : M, g.row_rec, c->stream, whole ? nullptr : idx));
                c->gen_obs_bf_valid = whole;
            }
            g.rows_idx = idx;
            g.rows_rec = reinterpret_cast<const float4*>(g.row_rec);
        } else if (two && c->gen_pre_idx == idx && c->gen_pre_M == M) {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_gather, 0));   // gathered ahead by the step before
        } else {
            HIPCHK(c, gather(idx, M, c->stream));
        }
        c->gen_pre_idx = nullptr;
        auto fork = [&]() -> hipError_t { const hipError_t e = hipEventRecord(c->ev_fork, c->stream); return e != hipSuccess ? e : hipStreamWaitEvent(c->stream2, c->ev_fork, 0); };
        auto join = [&]() -> hipError_t { const hipError_t e = hipEventRecord(c->ev_join, c->stream2); return e != hipSuccess ? e : hipStreamWaitEvent(c->stream, c->ev_join, 0); };
        hipStream_t s0 = two ? c->stream2 : c->stream;
        if (two && g.planes_dirty) { HIPCHK(c, gen_weight_planes(g, params, c->stream)); g.planes_dirty = false; }   // shared by both nets: before the fork
        if (two) HIPCHK(c, fork());
        const uint16_t* x0 = in_place ? g.obs_bf : g.xin_bf;
        // heads + loss + the head layers' backward in the forward launch's epilogue (kernels_generic_fused.hip: FusedLossArgs) where the shape allows
        g.head_fused = 0;
        const bool head_fused = paired && in_place && g.rows_rec != nullptr && !(c->cfg.kernel_flags & PPO_KERNEL_GENERIC_SPLIT_HEAD) && gen_fused_loss_ok(g, M);
        if (head_fused) {
            HIPCHK(c, gen_fused_forward_loss(g, params, x0, g.ld_in0, M, c->hp, 1.0 / global_M, global_M, c->cfg.norm_adv ? c->adv_stats + (size_t)slot * PPO_ADV_PARTS : nullptr,
                                             g.rows_idx, c->stream));
        } else if (paired) {
            HIPCHK(c, gen_fused_forward_both(g, params, x0, g.ld_in0, M, g.logits, g.val, c->stream, g.rows_idx));
        } else if (gen_fused_forward_ok(g) && M <= GEN_FUSED_MAX_ROWS) {   // bf16 storage: each net's forward pass is one launch that also leaves its hidden activations
            HIPCHK(c, gen_fused_forward(g, params, 1, nullptr, x0, g.ld_in0, M, true, g.logits, c->stream, g.rows_idx));
            HIPCHK(c, gen_fused_forward(g, params, 0, nullptr, x0, g.ld_in0, M, true, g.val, s0, g.rows_idx));
        } else {
            HIPCHK(c, gen_forward(g, params, 1, g.xin, M, g.acts[1], nullptr, nullptr, g.logits, c->stream));
            HIPCHK(c, gen_forward(g, params, 0, g.xin, M, g.acts[0], nullptr, nullptr, g.val, s0));
        }
        if (two) HIPCHK(c, join());   // the loss reads the logits and the values
        if (!head_fused) HIPCHK(c, gen_loss(GL, c->hp, g, M, 1.0 / global_M, global_M, c->cfg.norm_adv ? c->adv_stats + (size_t)slot * PPO_ADV_PARTS : nullptr, c->stream));
        // backward: paired too (A/B in one call, configs[4] share: 0.621 ms per optimizer step paired, 0.644 with the backward passes on two streams, 0.683 for
        // round 5's first form -- two streams throughout, gathered copies, four optimizer launches)
        if (two_bwd) HIPCHK(c, fork());
        c->gen_opt_fused_ok = false;
        if (paired_bwd) {
            HIPCHK(c, gen_backward_both(g, params, M, grads, c->stream));
        } else {
            hipStream_t sb = two_bwd ? c->stream2 : c->stream;
            HIPCHK(c, gen_backward(g, params, 1, g.xin, M, g.dlogits, grads, c->stream, two_bwd));
            HIPCHK(c, gen_backward(g, params, 0, g.xin, M, g.dval, grads, sb, two_bwd));
        }
        c->gen_opt_fused_ok = !classic && g.sq_valid[0] && g.sq_valid[1] && c->world == 1 && !c->force_collectives;   // nothing changes the gradient between the slab sums and the optimizer
        if (two_bwd) HIPCHK(c, join());   // the flat gradient is complete
        if (two && c->gen_next_idx && !in_place) {
            // nothing reads this step's gathered rows any more: the next step's gather (42 us of HBM streaming) runs on the second stream beside the
            // loss sums, the all-reduce, the norm, AdamW and the weight planes -- small kernels, one after the other, that leave the chip idle.
            // (Requested at the START of the step instead, into a second set of buffers, it does not overlap at all: its 16 k small workgroups take
            // every CU slot that frees up and the forward pass's one-per-CU workgroups start when it has drained -- measured, same time as no
            // gather-ahead.)
            HIPCHK(c, fork());
            HIPCHK(c, gather(c->gen_next_idx, c->gen_next_M, c->stream2));
            HIPCHK(c, hipEventRecord(c->ev_gather, c->stream2));
            c->gen_pre_idx = c->gen_next_idx; c->gen_pre_M = c->gen_next_M;
        }
    }
    if (!c->gen_opt_fused_ok) {   // (the fused optimizer launch adds the loss kernel's block sums itself)
        ProfScope ps(c, PROF_REDUCE);
        HIPCHK(c, gen_loss_sums(g, c->loss_sums, grads + GL.P, c->stream));
    }
    return PPO_OK;
}

// the T-step rollout as T x { actor GEMMs + heads, stores, env step } and one critic batch at the end (PPO_MultiDiscrete.cpp:547-575)
static ppo_status gen_rollout(ppo_ctx* c, const int64_t* forced) {
    GenericCtx& g = *c->gen;
    const GenLayout& GL = g.L;
    const int N = c->N;
    float* params = B_<float>(c, PPO_BUF_PARAMS);
    float* next_obs = B_<float>(c, PPO_BUF_NEXT_OBS);
    const uint8_t* mask = c->cfg.dist_kind == PPO_DIST_MASKED ? c->cur_mask : nullptr;
    if (gen_fused_ok(g)) {
        // bf16 storage: the T steps { observation, actor, heads, stores, env transition } in ONE launch, then the critic over all stored observations
        ProfScope ps(c, PROF_ROLLOUT);
        HIPCHK(c, gen_fused_rollout(g, params, c->cfg.dist_kind, N, c->T, c->cfg.max_episode_steps, c->cfg.seed, c->cfg.env_offset, c->rollout_steps,
                                    B_<int32_t>(c, PPO_BUF_EP_LEN), B_<float>(c, PPO_BUF_EP_REW), B_<float>(c, PPO_BUF_OBS), B_<uint8_t>(c, PPO_BUF_MASKS),
                                    B_<int32_t>(c, PPO_BUF_ACTIONS), B_<float>(c, PPO_BUF_LOGPROBS), B_<float>(c, PPO_BUF_REWARDS), B_<float>(c, PPO_BUF_DONES),
                                    B_<int32_t>(c, PPO_BUF_FIN_LEN), B_<float>(c, PPO_BUF_FIN_REW), next_obs, B_<int32_t>(c, PPO_BUF_NEXT_DONE), c->cur_mask, forced,
                                    c->stream));
        ppo_status s = gen_values(c, B_<float>(c, PPO_BUF_OBS), (int64_t)c->T * N, B_<float>(c, PPO_BUF_VALUES));
        if (s == PPO_OK) s = gen_values(c, next_obs, N, B_<float>(c, PPO_BUF_NEXT_VALUE));
        if (s != PPO_OK) return s;
    } else {
        ProfScope ps(c, PROF_ROLLOUT);
        for (int t = 0; t < c->T; t++) {
            const size_t tn = (size_t)t * N;
            const int64_t step = c->rollout_steps + t;
            HIPCHK(c, gen_forward(g, params, 1, next_obs, N, nullptr, g.dz[0], g.dz[1], g.logits, c->stream));
            HIPCHK(c, gen_heads(GL, c->cfg.dist_kind, g.logits, mask, forced ? forced + tn * GL.n_heads : nullptr, N, c->cfg.seed, c->cfg.env_offset, step,
                                g.act64, g.step_lp, g.step_en, c->stream));
            HIPCHK(c, gen_store_step(GL, N, next_obs, mask /* plain Categorical: masks are stored as all ones */, g.act64, g.step_lp, B_<int32_t>(c, PPO_BUF_NEXT_DONE), B_<float>(c, PPO_BUF_OBS) + tn * GL.obs,
                                     B_<uint8_t>(c, PPO_BUF_MASKS) + tn * GL.act, B_<int32_t>(c, PPO_BUF_ACTIONS) + tn * GL.n_heads,
                                     B_<float>(c, PPO_BUF_LOGPROBS) + tn, B_<float>(c, PPO_BUF_DONES) + tn, c->stream));
            HIPCHK(c, gen_synthetic_step(GL, N, c->cfg.seed, c->cfg.env_offset, step, c->cfg.max_episode_steps, B_<int32_t>(c, PPO_BUF_EP_LEN),
                                         B_<float>(c, PPO_BUF_EP_REW), next_obs, c->cur_mask, B_<float>(c, PPO_BUF_REWARDS) + tn,
                                         B_<int32_t>(c, PPO_BUF_NEXT_DONE), B_<int32_t>(c, PPO_BUF_FIN_LEN) + tn, B_<float>(c, PPO_BUF_FIN_REW) + tn, c->stream));
        }
        ppo_status s = gen_values(c, B_<float>(c, PPO_BUF_OBS), (int64_t)c->T * N, B_<float>(c, PPO_BUF_VALUES));
        if (s == PPO_OK) s = gen_values(c, next_obs, N, B_<float>(c, PPO_BUF_NEXT_VALUE));
        if (s != PPO_OK) return s;
    }
    c->rollout_steps += c->T;
    c->global_step += (int64_t)c->T * c->cfg.global_num_envs;
    c->fin_pending = true;
    return PPO_OK;
}

// The matrix-core update kernel gathers from one packed record per sample and net: (re)build them from the flattened rollout buffers
// (PPO_Discrete.cpp:557-562).  Also leaves the explained-variance partial sums (:647-648), which read the same returns / values.
static ppo_status pack_records(ppo_ctx* c) {
    if (!c->use_mfma || c->gen) return PPO_OK;
    // the observation's fp16 range is an error only where the wave-specialised matrix-core kernel will read the records: not with the one-wave kernel, and
    // not in an update whose weights send every launch to the vector kernel (the caller has refreshed the range snapshot: ppo_update, stand-alone step)
    const bool ws_will_run = !c->update_single_wave && weights_fit_update_mfma(c);
    HIPCHK(c, launch_pack_records(c->L, B_<float>(c, PPO_BUF_OBS), B_<int32_t>(c, PPO_BUF_ACTIONS),
                                  c->cfg.dist_kind == PPO_DIST_MASKED ? B_<uint8_t>(c, PPO_BUF_MASKS) : nullptr, B_<float>(c, PPO_BUF_LOGPROBS),
                                  B_<float>(c, PPO_BUF_ADVANTAGES), B_<float>(c, PPO_BUF_RETURNS), B_<float>(c, PPO_BUF_VALUES), c->B, c->rec_critic,
                                  c->rec_actor, c->ev_sums, ws_will_run ? c->error_flag : nullptr, c->stream));
    return PPO_OK;
}

static ppo_status fwd_bwd(ppo_ctx* c, const int32_t* idx, int64_t M, int slot, bool reduce = true) {
    if (c->gen) return gen_fwd_bwd(c, idx, M, slot);
    UpdateArgs a{};
    a.params = B_<float>(c, PPO_BUF_PARAMS);
    a.L = c->L;
    a.hp = c->hp;
    a.obs = B_<float>(c, PPO_BUF_OBS);
    a.actions = B_<int32_t>(c, PPO_BUF_ACTIONS);
    a.masks = c->cfg.dist_kind == PPO_DIST_MASKED ? B_<uint8_t>(c, PPO_BUF_MASKS) : nullptr;
    a.logprobs = B_<float>(c, PPO_BUF_LOGPROBS);
    a.advantages = B_<float>(c, PPO_BUF_ADVANTAGES);
    a.returns = B_<float>(c, PPO_BUF_RETURNS);
    a.values = B_<float>(c, PPO_BUF_VALUES);
    a.rec_critic = c->rec_critic;
    a.rec_actor = c->rec_actor;
    a.idx = idx;
    a.M = (int)M;
    a.global_M = (double)M * c->world;
    a.inv_global_M = 1.0 / a.global_M;
    c->last_global_M = a.global_M;
    a.adv_stat = c->adv_stats + (size_t)slot * PPO_ADV_PARTS;
    a.adv_norm = c->adv_norm + slot;
    a.slab = c->slab;
    a.stat_slab = c->stat_slab;
    a.stamps = c->stamping ? c->stamps : nullptr;
    a.error_flag = c->error_flag;
    a.single_wave = c->update_single_wave ? 1 : 0;
    bool mfma_now = c->use_mfma;
    if (mfma_now) {   // fp16 range of the matrix-core kernels' operands: see refresh_weight_range
        const ppo_status rs = refresh_weight_range(c);
        if (rs != PPO_OK) return rs;
        if (!c->wr_in_update) wr_snapshot(c);   // a stand-alone step; inside ppo_update the snapshot was taken once, at its start
        if (!weights_fit_update_mfma(c)) { mfma_now = false; c->vector_fallback_launches += 1; }
    }
    if (mfma_now) update_blocks_mfma((int)M, a.n_blocks);
    else a.n_blocks[0] = a.n_blocks[1] = update_blocks_per_net((int)M);
    {
        c->prof_last_sampled = c->prof_every <= 1 || (c->prof_count++ % c->prof_every) == c->prof_every / 2;
        ProfScope ps(c, PROF_FWD_BWD, c->prof_last_sampled);
        if (mfma_now) HIPCHK(c, launch_minibatch_fwd_bwd_mfma(a, c->stream));
        else HIPCHK(c, launch_minibatch_fwd_bwd(a, c->stream));
    }
    c->last_n_blocks[0] = a.n_blocks[0]; c->last_n_blocks[1] = a.n_blocks[1];
    if (reduce) {
        ProfScope ps(c, PROF_REDUCE);
        HIPCHK(c, launch_reduce_grads(c->slab, c->stat_slab, a.n_blocks, c->L, B_<float>(c, PPO_BUF_GRADS), c->loss_sums, c->stream));
    }
    return PPO_OK;
}

// clip + AdamW (or, with do_step false, just the loss scalars and the gradient norm) on whichever parameter layout the context has
static hipError_t clip_adamw_any(ppo_ctx* c, int slot, double global_M, int world, bool do_step, double* clipfrac_accum) {
    if (c->gen && c->gen_opt_fused_ok) {
        if (do_step) c->gen_opt_fused_ok = false;   // its partial sums belong to the gradient the step consumes
        return gen_opt_fused(*c->gen, B_<float>(c, PPO_BUF_PARAMS), B_<float>(c, PPO_BUF_GRADS), B_<float>(c, PPO_BUF_EXP_AVG), B_<float>(c, PPO_BUF_EXP_AVG_SQ),
                             c->cfg.max_grad_norm, c->adam_coefs + slot, c->loss_sums, global_M, c->hp, do_step, c->step_stats + slot, clipfrac_accum, c->error_flag, c->stream);
    }
    if (c->gen) {
        if (do_step) c->gen->planes_dirty = true;
        return gen_clip_adamw(B_<float>(c, PPO_BUF_PARAMS), B_<float>(c, PPO_BUF_GRADS), B_<float>(c, PPO_BUF_EXP_AVG), B_<float>(c, PPO_BUF_EXP_AVG_SQ),
                              c->gen->L, c->cfg.max_grad_norm, c->adam_coefs + slot, c->loss_sums, global_M, c->hp, world, do_step,
                              c->step_stats + slot, clipfrac_accum, c->norm2, c->stream);
    }
    return launch_clip_adamw(B_<float>(c, PPO_BUF_PARAMS), B_<float>(c, PPO_BUF_GRADS), B_<float>(c, PPO_BUF_EXP_AVG), B_<float>(c, PPO_BUF_EXP_AVG_SQ),
                             c->L, c->cfg.max_grad_norm, c->adam_coefs + slot, c->loss_sums, global_M, c->hp, world, do_step,
                             c->step_stats + slot, clipfrac_accum, c->norm2, c->stream, opt_guard(c));
}

extern "C" ppo_status ppo_minibatch_forward_backward(ppo_ctx* c, const int32_t* idx, int64_t M) {
    NEED(c, c && idx, "null argument");
    DeviceGuard dev_guard(c);
    NEED(c, M >= 1 && M <= c->B, "minibatch size out of range");
    const int slot = c->steps_per_update;  // scratch slot
    {   // stand-alone call: the caller may have rewritten any rollout buffer since the last pack
        if (c->use_mfma && !c->gen) {
            const ppo_status rs = refresh_weight_range(c);
            if (rs != PPO_OK) return rs;
            if (!c->wr_in_update) wr_snapshot(c);
        }
        const ppo_status ps = pack_records(c);
        if (ps != PPO_OK) return ps;
    }
    if (c->cfg.norm_adv) {
        HIPCHK(c, launch_adv_stats(B_<float>(c, PPO_BUF_ADVANTAGES), idx, M, M, 1, c->adv_stats + (size_t)slot * PPO_ADV_PARTS, c->stream));
        ppo_status s = allreduce_sum(c, c->adv_stats + (size_t)slot * PPO_ADV_PARTS, 2 * PPO_ADV_PARTS, true);
        if (s != PPO_OK) return s;
        HIPCHK(c, launch_adv_norm(c->adv_stats + (size_t)slot * PPO_ADV_PARTS, 1, 1, c->B, c->MB, M, c->world, c->adv_norm + slot, c->stream));
    }
    ppo_status s = fwd_bwd(c, idx, M, slot);
    if (s != PPO_OK) return s;
    // loss scalars and the norm of the unclipped gradient, without touching parameters
    HIPCHK(c, clip_adamw_any(c, slot, (double)M * c->world, 1, false, nullptr));
    c->last_stat_slot = slot;
    return PPO_OK;
}

extern "C" ppo_status ppo_allreduce_grads(ppo_ctx* c) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    if (c->world <= 1 && !c->force_collectives) return PPO_OK;
    // the slab reduction already left float copies of the loss sums behind the gradient: one collective carries both
    return allreduce_sum(c, c->buf[PPO_BUF_GRADS], (size_t)c->L.P + 8, false);
}

static ppo_status optimizer_step_slot(ppo_ctx* c, int slot, double global_M, bool coef_on_device) {
    c->opt_step += 1;
    if (!coef_on_device) {
        c->adam_coefs_h[slot] = adam_coef(c->lr, c->opt_step);
        HIPCHK(c, hipMemcpyAsync(c->adam_coefs + slot, c->adam_coefs_h + slot, sizeof(AdamCoef), hipMemcpyHostToDevice, c->stream));
    }
    {
        ProfScope ps(c, PROF_OPT);
        HIPCHK(c, clip_adamw_any(c, slot, global_M, c->force_collectives ? 2 : c->world /* self-test: read the loss sums from the all-reduced tail */, true,
                                 c->clipfrac_accum));
    }
    c->last_stat_slot = slot;
    return PPO_OK;
}

extern "C" ppo_status ppo_optimizer_step(ppo_ctx* c) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    // stand-alone use: the slot's pinned coefficient must not be rewritten while a previous copy is in flight
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const ppo_status s = optimizer_step_slot(c, c->steps_per_update, c->last_global_M, false);
    return s != PPO_OK ? s : sweep_weight_range(c, c->stream);
}

extern "C" ppo_status ppo_set_learning_rate(ppo_ctx* c, double lr) {
    NEED(c, c != nullptr, "null ctx");
    c->lr = lr;
    return PPO_OK;
}

// All epochs x minibatches of one update, PPO_Discrete.cpp:567-644, then explained variance (:647-648).
extern "C" ppo_status ppo_update(ppo_ctx* c) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    const int E = c->cfg.update_epochs, nmb = c->n_mb;
    ppo_status s = PPO_OK;
    const int32_t* perm = B_<int32_t>(c, PPO_BUF_PERM);
    // returns, values and advantages are fixed for the whole update: the sample records and the explained-variance sums (:647-648) are formed first,
    // so that a sharded run can send its statistics along with the advantage sums
    if (!c->gen) {   // the fp16-range snapshot of this update first: pack_records asks it whether the matrix-core kernel will read the records
        s = refresh_weight_range(c);
        if (s == PPO_OK) s = sweep_weight_range(c, c->wr_stream);
        if (s != PPO_OK) return s;
        wr_snapshot(c);
    }
    s = pack_records(c);
    if (s != PPO_OK) return s;
    struct InUpdate {
        ppo_ctx* c;
        explicit InUpdate(ppo_ctx* x) : c(x) { c->wr_in_update = true; c->gen_obs_bf_valid = false; }
        ~InUpdate() { c->wr_in_update = false; c->gen_obs_bf_valid = false; }
    } in_update(c);
    if (!c->use_mfma || c->gen)   // otherwise pack_records left the sums
        HIPCHK(c, launch_explained_variance(B_<float>(c, PPO_BUF_RETURNS), B_<float>(c, PPO_BUF_VALUES), c->B, c->ev_sums, c->stream));
    // sharded: this rank's slot of the job-global statistics block (explained-variance sums, its ring of finished episodes with their positions in
    // the reference's push order) rides in front of the advantage sums -- one all-reduce per update, and ppo_read_stats is the same on every rank
    const bool sharded = (c->world > 1 || c->force_collectives) && c->world <= 8;
    if (sharded) {
        s = consume_finished_episodes(c);
        if (s != PPO_OK) return s;
        HIPCHK(c, launch_gstats_pack(c->ev_sums, c->ring, c->rank, c->gstats, c->stream));
    }
    if (c->cfg.norm_adv) {
        // the advantages and the permutations are fixed for the whole update: the permutations of all epochs and the statistics of ALL
        // minibatches in one launch (and, when sharded, one small all-reduce) instead of a reduction inside every optimizer step
        HIPCHK(c, launch_permutations_adv_stats(B_<float>(c, PPO_BUF_ADVANTAGES), B_<int32_t>(c, PPO_BUF_PERM), c->B, E, c->MB, c->cfg.seed, c->updates,
                                                c->rank, c->adv_stats, c->stream));
        if (sharded) s = allreduce_sum(c, c->gstats, (size_t)PPO_GSTAT_DOUBLES + (size_t)2 * E * nmb * PPO_ADV_PARTS, true);
        else s = allreduce_sum(c, c->adv_stats, (size_t)2 * E * nmb * PPO_ADV_PARTS, true);
        if (s != PPO_OK) return s;
        HIPCHK(c, launch_adv_norm(c->adv_stats, E * nmb, nmb, c->B, c->MB, 0, c->world, c->adv_norm, c->stream, c->clipfrac_accum));   // also: m_clipfracs reset, :564
    } else {
        s = ppo_generate_permutations(c);
        if (s != PPO_OK) return s;
        if (sharded) {
            s = allreduce_sum(c, c->gstats, (size_t)PPO_GSTAT_DOUBLES, true);
            if (s != PPO_OK) return s;
        }
    }
    c->have_gstats = sharded;
    // AdamW scalars of every step of this update, one async copy.  The pinned mirror has two halves used alternately; a half is
    // rewritten only once its previous copy has completed (normally long ago: no stall, and no stream-wide synchronisation).
    {
        const int half = c->coef_half;
        c->coef_half ^= 1;
        AdamCoef* h = c->adam_coefs_h + (size_t)half * (c->steps_per_update + 1);
        HIPCHK(c, hipEventSynchronize(c->coef_copied[half]));
        for (int k = 0; k < E * nmb; k++) h[k] = adam_coef(c->lr, c->opt_step + 1 + k);
        HIPCHK(c, hipMemcpyAsync(c->adam_coefs, h, (size_t)E * nmb * sizeof(AdamCoef), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->coef_copied[half], c->stream));
    }
    if (!c->cfg.norm_adv) HIPCHK(c, hipMemsetAsync(c->clipfrac_accum, 0, 2 * sizeof(double), c->stream));  // m_clipfracs reset, :564 (with norm_adv: cleared by the adv_norm launch above)
    c->gen_pre_idx = nullptr;   // nothing gathered ahead survives an update (an error return may have left a claim behind)
    int k = 0;
    for (int e = 0; e < E; e++) {
        for (int mbi = 0; mbi < nmb; mbi++, k++) {
            const int64_t start = (int64_t)mbi * c->MB;
            const int64_t M = std::min<int64_t>(c->MB, c->B - start);
            const bool fused = c->world == 1 && !c->force_collectives && !c->gen;
            // sharded on the direct-exchange transport: the same two launches, the exchange folded into the slab reduction
            const bool fused_x = c->world > 1 && c->xchg && !c->force_collectives && !c->gen;
            {   // the step after this one (next minibatch, or the first one of the next epoch), for the generic path's gather-ahead
                const bool more = mbi + 1 < nmb || e + 1 < E;
                const int e2 = mbi + 1 < nmb ? e : e + 1, m2 = mbi + 1 < nmb ? mbi + 1 : 0;
                const int64_t start2 = (int64_t)m2 * c->MB;
                c->gen_next_idx = more ? perm + (size_t)e2 * c->B + start2 : nullptr;
                c->gen_next_M = more ? std::min<int64_t>(c->MB, c->B - start2) : 0;
            }
            s = fwd_bwd(c, perm + (size_t)e * c->B + start, M, k, !(fused || fused_x));
            c->gen_next_idx = nullptr;
            if (s != PPO_OK) return s;
            if (fused_x) {
                ExchangeComm& x = *c->xchg;
                c->opt_step += 1;
                x.seq += 1;
                ProfScope ps(c, PROF_OPT);
                HIPCHK(c, launch_reduce_exchange_clip_adamw(c->slab, c->stat_slab, c->last_n_blocks, c->L, B_<float>(c, PPO_BUF_GRADS), c->loss_sums,
                                                            B_<float>(c, PPO_BUF_PARAMS), B_<float>(c, PPO_BUF_EXP_AVG), B_<float>(c, PPO_BUF_EXP_AVG_SQ),
                                                            c->cfg.max_grad_norm, c->adam_coefs + k, (double)M * c->world, c->hp, c->step_stats + k,
                                                            c->clipfrac_accum, c->fused_partial, x.peer, c->rank, c->world, x.slot_bytes, x.seq, x.timeout_flag,
                                                            c->stream, opt_guard(c)));
                c->last_stat_slot = k;
                continue;
            }
            if (fused) {
                // single rank: two launches (slab reduction + sums of squares, then norm + clip + AdamW) instead of three
                c->opt_step += 1;
                ProfScope ps(c, PROF_OPT);
                HIPCHK(c, launch_reduce_clip_adamw(c->slab, c->stat_slab, c->last_n_blocks, c->L, B_<float>(c, PPO_BUF_GRADS), c->loss_sums, B_<float>(c, PPO_BUF_PARAMS),
                                                   B_<float>(c, PPO_BUF_EXP_AVG), B_<float>(c, PPO_BUF_EXP_AVG_SQ), c->cfg.max_grad_norm, c->adam_coefs + k, (double)M,
                                                   c->hp, c->step_stats + k, c->clipfrac_accum, c->fused_partial, c->stream, opt_guard(c)));
                c->last_stat_slot = k;
                continue;
            }
            s = ppo_allreduce_grads(c);
            if (s != PPO_OK) return s;
            s = optimizer_step_slot(c, k, (double)M * c->world, true);
            if (s != PPO_OK) return s;
        }
    }
    c->have_ev = true;
    c->updates += 1;
    return PPO_OK;
}

// One iteration of PPO_Discrete::train()'s loop (:511-659) without printing / checkpointing.
extern "C" ppo_status ppo_train_iteration(ppo_ctx* c) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    if (c->cfg.anneal_lr && c->num_updates_total > 0) {
        // frac = 1.0 - (update - 1.0) / num_updates; lr_now = frac * m_learning_rate  (:515-517), update is 1-based
        const double frac = 1.0 - ((double)(c->updates + 1) - 1.0) / (double)c->num_updates_total;
        c->lr = frac * c->cfg.learning_rate;
    }
    ppo_status s = ppo_rollout(c, nullptr);
    if (s != PPO_OK) return s;
    // NEXT_VALUE was produced by the rollout's epilogue with the same parameters: go straight to the scan
    s = run_scan(c);
    if (s != PPO_OK) return s;
    return ppo_update(c);
}

// Statistics are read in two steps so that reading them does not have to drain the stream:
//   ppo_stats_snapshot       enqueues, behind everything enqueued so far, asynchronous copies of every device piece the statistics are made of into
//                            ONE pinned host block, records an event, and notes the host-side training state (learning rate, step counters);
//   ppo_stats_snapshot_read  waits for THAT event (not for the stream) and decodes the block.
// A host that prints a table per update (the facade's train()) takes the snapshot right behind ppo_update, enqueues the next iteration, and only then
// reads: the GPU works through iteration k + 1 while the host formats iteration k.  ppo_read_stats = snapshot + read.
extern "C" ppo_status ppo_stats_snapshot(ppo_ctx* c) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    ppo_status s = consume_finished_episodes(c);
    if (s != PPO_OK) return s;
    if (c->snap_count == 2) return fail(c, PPO_ERR_STATE, "two statistics snapshots are pending: read one (ppo_stats_snapshot_read) before taking a third");
    const int slot = (c->snap_oldest + c->snap_count) & 1;
    StatsSnap* h = c->snap + slot;
    h->have_step = c->last_stat_slot >= 0;
    h->have_ev = c->have_ev;
    h->have_gstats = c->have_gstats;
    h->world = c->world; h->B = c->B; h->global_step = c->global_step; h->opt_step = c->opt_step; h->updates = c->updates; h->lr = c->lr;
    h->xchg_flag = 0;
    HIPCHK(c, hipMemcpyAsync(&h->error_flag, c->error_flag, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (c->xchg && c->xchg->timeout_flag) HIPCHK(c, hipMemcpyAsync(&h->xchg_flag, c->xchg->timeout_flag, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (h->have_step) HIPCHK(c, hipMemcpyAsync(&h->st, c->step_stats + c->last_stat_slot, sizeof h->st, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h->cf, c->clipfrac_accum, sizeof h->cf, hipMemcpyDeviceToHost, c->stream));
    if (h->have_gstats) HIPCHK(c, hipMemcpyAsync(h->gstats, c->gstats, sizeof h->gstats, hipMemcpyDeviceToHost, c->stream));
    else {
        if (h->have_ev) HIPCHK(c, hipMemcpyAsync(h->ev, c->ev_sums, sizeof h->ev, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&h->ring, c->ring, sizeof h->ring, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipEventRecord(c->snap_ev[slot], c->stream));
    c->snap_count += 1;
    return PPO_OK;
}

extern "C" ppo_status ppo_stats_snapshot_read(ppo_ctx* c, ppo_stats* out) {
    NEED(c, c && out, "null argument");
    DeviceGuard dev_guard(c);
    NEED(c, c->snap_count > 0, "no statistics snapshot is pending (ppo_stats_snapshot)");
    std::memset(out, 0, sizeof *out);
    const int slot = c->snap_oldest;
    HIPCHK(c, hipEventSynchronize(c->snap_ev[slot]));
    c->snap_oldest ^= 1;
    c->snap_count -= 1;
    const StatsSnap& h = c->snap[slot];
    if (h.error_flag & 1) return fail(c, PPO_ERR_STATE, "CartPole reset-stream table exhausted (capacity %d resets per env)", c->reset_cap);
    if (h.error_flag & PPO_ERRFLAG_ROLLOUT_RANGE)
        return fail(c, PPO_ERR_STATE, "rollout: an output-layer weight of the actor is >= 255 in magnitude and does not fit the fp16 operand of the matrix-core rollout "
                                      "(its logits are invalid): create the context with PPO_KERNEL_ROLLOUT_VECTOR in ppo_config.kernel_flags");
    if (h.error_flag & PPO_ERRFLAG_UPDATE_PROTOCOL)
        return fail(c, PPO_ERR_STATE, "update kernel: a bounded wait between its forward and gradient waves ran out (the gradient of that step was incomplete: "
                                      "parameters are undefined from there on)");
    if (h.error_flag & PPO_ERRFLAG_GAE_PROTOCOL)
        return fail(c, PPO_ERR_STATE, "advantage scan: a bounded wait between the mover waves and the walker of the time-pipelined kernel ran out (the advantages and "
                                      "returns of that strip of envs are NaN, and so is everything trained on them)");
    if (h.error_flag & PPO_ERRFLAG_UPDATE_RANGE)
        return fail(c, PPO_ERR_STATE, "update kernel: an observation of magnitude >= 65504 does not fit the fp16 operand of the matrix-core update kernel; the optimizer "
                                      "gradient of that update is not finite and the parameters are undefined from there on: create the context with "
                                      "PPO_KERNEL_UPDATE_VECTOR in ppo_config.kernel_flags");
    if (h.xchg_flag != 0)
        return fail(c, PPO_ERR_COMM, "direct exchange: an all-reduce gave up waiting for a peer after %.1f s; its sums were incomplete, the replicas have "
                                     "diverged and the communicator is dead (every later all-reduce returns at once)", c->xchg ? c->xchg->wait_seconds : 0.0);
    if (h.have_step) {
        const StepStats& st = h.st;
        out->pg_loss = st.pg_loss; out->v_loss = st.v_loss; out->entropy_loss = st.entropy_loss; out->approx_kl = st.approx_kl;
        out->loss = st.loss; out->clipfrac_last = st.clipfrac; out->total_norm = st.total_norm;
    }
    out->clipfrac_mean = h.cf[1] > 0 ? h.cf[0] / h.cf[1] : 0.0;
    if (h.have_gstats) {
        // sharded: the all-reduced block of the last ppo_update holds every rank's explained-variance sums and ring -- the same bytes on every rank
        const double* g = h.gstats;
        const double n = (double)h.B * h.world;
        const double var_y = (g[1] - g[0] * g[0] / n) / (n - 1.0), var_d = (g[3] - g[2] * g[2] / n) / (n - 1.0);
        out->explained_variance = (double)(1.0f - (float)var_d / (float)var_y);  // :647-648 over the job's whole batch
        // the job's CircularBuffer(100): the newest 100 episodes of the union, in the reference's push order (step, then global env index)
        struct Ep { double key, len, rew; };
        std::vector<Ep> eps;
        double total = 0;
        for (int r = 0; r < h.world; r++) {
            const double* slot = g + PPO_GSTAT_HEAD + (size_t)r * PPO_GSTAT_RANK;
            total += slot[0];
            const int size = (int)slot[1];
            for (int i = 0; i < size && i < 100; i++) eps.push_back({ slot[4 + 3 * i], slot[4 + 3 * i + 1], slot[4 + 3 * i + 2] });
        }
        std::sort(eps.begin(), eps.end(), [](const Ep& a, const Ep& b) { return a.key < b.key; });
        const size_t keep = std::min<size_t>(eps.size(), 100), first = eps.size() - keep;
        if (keep > 0) {
            // the reference sums the ring from slot 0 upwards (Utils.h:72-78): lay the merged episodes out as the single ring would hold them --
            // episode number j of the job sits in slot j % 100
            std::vector<Ep> ring_order(keep);
            for (size_t i = 0; i < keep; i++) {
                const long long j = (long long)total - (long long)keep + (long long)i;   // 0-based episode number
                ring_order[keep < 100 ? i : (size_t)(j % 100)] = eps[first + i];
            }
            double sl = 0, sr = 0;
            for (size_t i = 0; i < keep; i++) { sl += ring_order[i].len; sr += ring_order[i].rew; }
            out->ep_len_mean = sl / (double)keep;
            out->ep_rew_mean = (double)(float)(sr / (double)keep);
        }
        out->ep_count = (int32_t)keep;
    } else {
        if (h.have_ev) {
            double sy = 0, sy2 = 0, sd = 0, sd2 = 0;
            for (int b = 0; b < PPO_EV_BLOCKS; b++) { sy += h.ev[b * 4]; sy2 += h.ev[b * 4 + 1]; sd += h.ev[b * 4 + 2]; sd2 += h.ev[b * 4 + 3]; }
            const double n = (double)h.B;
            const double var_y = (sy2 - sy * sy / n) / (n - 1.0), var_d = (sd2 - sd * sd / n) / (n - 1.0);
            out->explained_variance = (double)(1.0f - (float)var_d / (float)var_y);  // :647-648 (float tensors)
        }
        const EpisodeRing& ring = h.ring;
        if (ring.size > 0) {
            double sl = 0, sr = 0;
            for (int i = 0; i < ring.size; i++) { sl += ring.len[i]; sr += ring.rew[i]; }
            out->ep_len_mean = sl / ring.size;                 // CircularBuffer::avgLength (Utils.h:76-78)
            out->ep_rew_mean = (double)(float)(sr / ring.size); // avgReward returns float (Utils.h:72-74)
        }
        out->ep_count = ring.size;
    }
    out->learning_rate = h.lr;
    out->global_step = h.global_step;
    out->optimizer_steps = h.opt_step;
    out->updates = h.updates;
    return PPO_OK;
}

extern "C" ppo_status ppo_read_stats(ppo_ctx* c, ppo_stats* out) {
    NEED(c, c && out, "null argument");
    NEED(c, c->snap_count == 0, "a statistics snapshot is pending: read it (ppo_stats_snapshot_read) before ppo_read_stats");
    ppo_status s = ppo_stats_snapshot(c);
    if (s != PPO_OK) return s;
    s = ppo_stats_snapshot_read(c, out);
    if (s != PPO_OK) return s;
    // callers rely on ppo_read_stats leaving the stream idle (it always synchronised)
    DeviceGuard dev_guard(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PPO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Measurement
// ---------------------------------------------------------------------------------------------------------
extern "C" ppo_status ppo_profile_enable(ppo_ctx* c, int32_t on) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    // on: 0 = off, 1 = every instrumented launch, 2 = only the dominant kernel (fwd/bwd; one launch in 8) and the GAE scan,
    //     3 = in-kernel phase stamps of the dominant kernel (diagnostic variant: read its SHARES, never its run time),
    //     4 = as 2 with one launch in 41 (about one per update, a different step of the update every time)
    const bool sampled = on == 2 || on == 4;
    c->profiling = (on == 0 || on == 3) ? 0u : (sampled ? ((1u << PROF_FWD_BWD) | (1u << PROF_GAE) | (1u << PROF_ALLREDUCE)) : 0xffffffffu);
    c->prof_every = on == 2 ? 8 : (on == 4 ? 41 : 1);   // modes 2 / 4 sample the update kernel (5 / ~1 of an update's 40 launches), every GAE launch
    c->prof_count = 0;
    c->stamping = on == 3;
    if (c->profiling) {   // events are created here, not inside the region being timed
        while (c->event_pool.size() < 256) {
            hipEvent_t e = nullptr;
            HIPCHK(c, hipEventCreate(&e));
            c->event_pool.push_back(e);
        }
    }
    if (c->stamping) HIPCHK(c, hipMemsetAsync(c->stamps, 0, 24 * sizeof(unsigned long long), c->stream));
    return PPO_OK;
}

extern "C" ppo_status ppo_profile_read(ppo_ctx* c, ppo_profile* out) {
    NEED(c, c && out, "null argument");
    DeviceGuard dev_guard(c);
    std::memset(out, 0, sizeof *out);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {
        const ppo_status hs = comm_health(c);
        if (hs != PPO_OK) return hs;
    }
    int64_t* cnt[PROF_KINDS_] = { &out->fwd_bwd_launches, &out->gae_launches, &out->rollout_launches, &out->optimizer_launches, &out->reduce_launches, &out->allreduce_launches };
    double* ms[PROF_KINDS_] = { &out->fwd_bwd_ms, &out->gae_ms, &out->rollout_ms, &out->optimizer_ms, &out->reduce_ms, &out->allreduce_ms };
    for (auto& sp : c->spans) {
        float t = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&t, sp.a, sp.b));
        *cnt[sp.kind] += 1;
        *ms[sp.kind] += (double)t;
        c->event_pool.push_back(sp.a);
        c->event_pool.push_back(sp.b);
    }
    c->spans.clear();
    unsigned long long st[24];
    HIPCHK(c, hipMemcpy(st, c->stamps, sizeof st, hipMemcpyDeviceToHost));
    for (int i = 0; i < 24; i++) out->phase_cycles[i] = (double)st[i];
    out->vector_fallback_launches = c->vector_fallback_launches;
    return PPO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Multi-GPU
// ---------------------------------------------------------------------------------------------------------
extern "C" ppo_status ppo_comm_unique_id(void* id_out_h) {
    if (!id_out_h) return PPO_ERR_INVALID;
    std::string err;
    if (!rccl::load(err)) { g_create_error = err; return PPO_ERR_COMM; }
    rccl::UniqueId id;
    std::memset(&id, 0, sizeof id);
    const int rc = rccl::GetUniqueId(&id);
    if (rc != 0) { g_create_error = "ncclGetUniqueId failed"; return PPO_ERR_COMM; }
    static_assert(sizeof(rccl::UniqueId) == PPO_COMM_ID_BYTES, "unique id size");
    std::memcpy(id_out_h, &id, sizeof id);
    return PPO_OK;
}

extern "C" ppo_status ppo_comm_init(ppo_ctx* c, const void* id_h, int32_t rank, int32_t nranks) {
    NEED(c, c && id_h, "null argument");
    DeviceGuard dev_guard(c);
    NEED(c, nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
    NEED(c, nranks <= 8, "more than 8 ranks: the job-global statistics block (ppo_read_stats) and the transports serve the 8 GPUs of one node");
    NEED(c, c->cfg.global_num_envs == (int64_t)c->cfg.num_envs * nranks, "global_num_envs must equal num_envs * nranks (equal shards)");
    // ppo_config.kernel_flags & PPO_KERNEL_COMM_SELFTEST: a ONE-rank communicator is really created and every collective of the multi-rank path is
    // really issued (sums over one rank = identity): the RCCL calls on a box with a single GPU
    const bool selftest = nranks == 1 && (c->cfg.kernel_flags & PPO_KERNEL_COMM_SELFTEST) != 0;
    if (nranks == 1 && !selftest) { c->world = 1; c->rank = 0; return PPO_OK; }
    std::string err;
    if (!rccl::load(err)) return fail(c, PPO_ERR_COMM, "%s", err.c_str());
    rccl::UniqueId id;
    std::memcpy(&id, id_h, sizeof id);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int rc = rccl::CommInitRank(&c->comm, nranks, id, rank);
    if (rc != 0) return fail(c, PPO_ERR_COMM, "ncclCommInitRank failed: %s", rccl::GetErrorString ? rccl::GetErrorString(rc) : "?");
    c->world = nranks;
    c->rank = rank;
    c->force_collectives = selftest;
    return PPO_OK;
}

// Joins the in-process group `group_id` as rank `rank` of `nranks` (all members live in this process, one host thread each).
extern "C" ppo_status ppo_comm_init_local(ppo_ctx* c, int64_t group_id, int32_t rank, int32_t nranks) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    NEED(c, nranks >= 1 && nranks <= 8 && rank >= 0 && rank < nranks, "bad rank / nranks (in-process groups hold at most 8 contexts)");
    NEED(c, c->cfg.global_num_envs == (int64_t)c->cfg.num_envs * nranks, "global_num_envs must equal num_envs * nranks (equal shards)");
    NEED(c, c->comm == nullptr && !c->lgroup, "context already has a communicator");
    if (nranks == 1) { c->world = 1; c->rank = 0; return PPO_OK; }
    std::shared_ptr<LocalGroup> g;
    {
        std::lock_guard<std::mutex> lk(g_local_groups_mu);
        auto it = g_local_groups.find(group_id);
        if (it == g_local_groups.end()) {
            g = std::make_shared<LocalGroup>();
            g->n = nranks;
            g->device = c->cfg.device;
            HIPCHK(c, hipSetDevice(c->cfg.device));   // the group's events belong to its device
            HIPCHK(c, hipEventCreateWithFlags(&g->done[0], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&g->done[1], hipEventDisableTiming));
            g_local_groups[group_id] = g;
        } else {
            g = it->second;
        }
        NEED(c, g->n == nranks, "group was created with a different size");
        // one kernel on one device reads every member's buffer and the members wait on shared events: all of that is per device.  Several GPUs
        // are driven by one process per GPU (ppo_comm_init / ppo_comm_init_exchange)
        NEED(c, g->device == c->cfg.device, "in-process groups hold contexts of ONE device; use one process per GPU for several");
        if (++g->joined == g->n) g_local_groups.erase(group_id);  // complete: the id may be reused by a later group
    }
    HIPCHK(c, hipEventCreateWithFlags(&c->lg_ready, hipEventDisableTiming));
    c->lgroup = g;
    c->world = nranks;
    c->rank = rank;
    return PPO_OK;
}

// ---- one-shot direct exchange (see ExchangeComm) ----
// Step 1 (every rank): allocate the exchange buffer and export its IPC handle.  payload capacity: the larger of the gradient (+ 8 loss sums) and the
// advantage sums of one update.
extern "C" ppo_status ppo_comm_exchange_handle(ppo_ctx* c, void* handle_out_h) {
    NEED(c, c && handle_out_h, "null argument");
    DeviceGuard dev_guard(c);
    NEED(c, c->comm == nullptr && !c->lgroup, "context already has a communicator");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (!c->xchg) {
        std::unique_ptr<ExchangeComm> x(new ExchangeComm());
        const size_t grad = ((size_t)c->L.P + 8) * sizeof(float), adv = ((size_t)PPO_GSTAT_DOUBLES + (size_t)2 * c->steps_per_update * PPO_ADV_PARTS) * sizeof(double);   // [statistics block | advantage sums]
        x->slot_bytes = (std::max(grad, adv) + 255) / 256 * 256;
        const size_t total = 16 * x->slot_bytes + 256;   // [2 parities][8 source ranks] payload slots, 16 flags, 2 counters (kernels_update.hip: xchg_slot)
        // fine-grained device memory: stores of a running kernel become visible to peers' running kernels (coarse-grained memory is only
        // coherent at kernel boundaries); plain hipMalloc as a fallback on a single device
        hipError_t e = hipExtMallocWithFlags(&x->own, total, hipDeviceMallocFinegrained);
        if (e != hipSuccess) { (void)hipGetLastError(); HIPCHK(c, hipMalloc(&x->own, total)); }
        HIPCHK(c, hipMemset(x->own, 0, total));
        HIPCHK(c, dalloc(c, &x->timeout_flag, 4));
        const unsigned long long ticks = (unsigned long long)(x->wait_seconds * 1e8);   // s_memrealtime: 100 MHz
        HIPCHK(c, hipMemcpy(x->timeout_flag + 2, &ticks, sizeof ticks, hipMemcpyHostToDevice));
        HIPCHK(c, hipDeviceSynchronize());
        c->xchg = std::move(x);
    }
    static_assert(sizeof(hipIpcMemHandle_t) <= PPO_COMM_HANDLE_BYTES, "IPC handle size");
    hipIpcMemHandle_t h;
    HIPCHK(c, hipIpcGetMemHandle(&h, c->xchg->own));
    std::memset(handle_out_h, 0, PPO_COMM_HANDLE_BYTES);
    std::memcpy(handle_out_h, &h, sizeof h);
    return PPO_OK;
}

// Step 2 (every rank, after all handles have been gathered, e.g. over torch.distributed): map the peers' buffers and switch the context's
// all-reduces to the exchange.  handles_h: nranks x PPO_COMM_HANDLE_BYTES in rank order.
extern "C" ppo_status ppo_comm_init_exchange(ppo_ctx* c, const void* handles_h, int32_t rank, int32_t nranks) {
    NEED(c, c && handles_h, "null argument");
    DeviceGuard dev_guard(c);
    NEED(c, nranks >= 1 && nranks <= 8 && rank >= 0 && rank < nranks, "bad rank / nranks (the direct exchange serves the 8 GPUs of one node)");
    NEED(c, c->cfg.global_num_envs == (int64_t)c->cfg.num_envs * nranks, "global_num_envs must equal num_envs * nranks (equal shards)");
    NEED(c, c->xchg && c->xchg->own, "call ppo_comm_exchange_handle first");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    ExchangeComm& x = *c->xchg;
    for (int r = 0; r < nranks; r++) {
        if (r == rank) { x.peer[r] = x.own; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const char*>(handles_h) + (size_t)r * PPO_COMM_HANDLE_BYTES, sizeof h);
        HIPCHK(c, hipIpcOpenMemHandle(&x.peer[r], h, hipIpcMemLazyEnablePeerAccess));
        x.opened[r] = true;
    }
    c->world = nranks;
    c->rank = rank;
    c->force_collectives = nranks == 1;   // a one-rank exchange still runs the multi-rank code path (self-test)
    return PPO_OK;
}

// How long a kernel of the direct exchange waits for a peer's share before it gives up (default 30 s).  Call after ppo_comm_exchange_handle.
extern "C" ppo_status ppo_comm_set_wait_limit(ppo_ctx* c, double seconds) {
    NEED(c, c != nullptr, "null ctx");
    DeviceGuard dev_guard(c);
    NEED(c, c->xchg && c->xchg->timeout_flag, "call ppo_comm_exchange_handle first");
    NEED(c, seconds >= 1e-3 && seconds <= 3600.0, "wait limit must lie in [1 ms, 1 h]");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->xchg->wait_seconds = seconds;
    const unsigned long long ticks = (unsigned long long)(seconds * 1e8);
    HIPCHK(c, hipMemcpy(c->xchg->timeout_flag + 2, &ticks, sizeof ticks, hipMemcpyHostToDevice));
    return PPO_OK;
}

// 0: no all-reduce kernel of this context has given up waiting for a peer; non-zero: at least one has (a peer died or never joined) -- read it as
// a boolean: every polling lane that gives up adds to it, and so does every later call on the dead communicator.  Never fails because of the
// flag itself (ppo_sync / ppo_read_stats / ppo_profile_read do: PPO_ERR_COMM).
extern "C" ppo_status ppo_comm_exchange_timeouts(ppo_ctx* c, int32_t* count_out) {
    NEED(c, c && count_out, "null argument");
    DeviceGuard dev_guard(c);
    *count_out = 0;
    if (!c->xchg) return PPO_OK;
    HIPCHK(c, hipMemcpyAsync(count_out, c->xchg->timeout_flag, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PPO_OK;
}
