// pg_api.cpp -- host side of the C ABI declared in include/pindel_pg.h.
// Owns the GPU context, packs the reference into bit planes, moves read batches
// to HBM, launches the search kernel (pg_kernels.hip) and turns its pooled runs
// back into per-read CSR results.  There is NO CPU implementation of the search
// here: without a working HIP device every entry point fails with PG_E_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pg_device.h"
#include "pindel_pg.h"

#include <chrono>
#include <condition_variable>
#include <functional>
#include <exception>
#include <pthread.h>
namespace {

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
const bool g_host_timing = getenv("PG_HOST_TIMING") != nullptr;

// The environment switches of tests and experiments, read ONCE (getenv on the 50 000-read flush path is a linear scan of the
// environment per call) -- pg_debug_reload_env() re-reads them (tests that change one in mid-process call it).  g_env is a plain
// object read without synchronisation by every caller's thread and by the host pool's workers: pg_debug_reload_env() is a TEST
// hook and must only be called while no pg_* call is in flight on any thread (pindel_pg.h says so; the tests do).
PgEnvSwitches g_env;
std::once_flag g_env_once;
void load_env()
{
    PgEnvSwitches e;
    const char *c = getenv("PG_HOST_CHUNK");                 // (tests: several chunks on a small batch)
    e.host_chunk = c ? (uint32_t)std::min<long>(std::max(1, atoi(c)), PG_DELIVER_CHUNK) : 0u;
    e.no_single_block = getenv("PG_NO_SINGLE_BLOCK") != nullptr;
    e.tiny_delivery = getenv("PG_TEST_TINY_DELIVERY") != nullptr;
    e.tiny_pool = getenv("PG_TEST_TINY_POOL") != nullptr;
    e.force_wide_cells = getenv("PG_FORCE_WIDE_CELLS") != nullptr;
    e.split_launch = getenv("PG_SPLIT_LAUNCH") != nullptr;
    e.generic_kernels = getenv("PG_GENERIC_KERNELS") != nullptr;
    e.lds_pad = getenv("PG_LDS_PAD") ? (uint32_t)atoi(getenv("PG_LDS_PAD")) : 0u;
    e.no_pack_in_place = getenv("PG_NO_PACK_IN_PLACE") != nullptr;
    e.pack_claim = getenv("PG_PACK_CLAIM") ? (uint32_t)std::max(1, atoi(getenv("PG_PACK_CLAIM"))) : 0u;
    e.pack_in_place_min = getenv("PG_PACK_IN_PLACE_MIN") ? (uint32_t)std::max(1, atoi(getenv("PG_PACK_IN_PLACE_MIN"))) : PG_PACK_IN_PLACE_MIN;
    g_env = e;
}
const PgEnvSwitches &env()
{
    std::call_once(g_env_once, load_env);
    return g_env;
}

// Pinned host memory for results (device-to-host copies at PCIe speed, no staging through pageable pages).
// Pinning is expensive, results are handed out and freed all the time (Pindel flushes a batch every 50 000
// reads): freed blocks go to a small process-wide cache and are reused by the next result of similar size.
struct PinnedCache {
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> free_blocks;
    size_t cached_bytes = 0;
    // page-locked memory kept for reuse: 6 GiB unless PG_PINNED_CACHE_MB says otherwise (0 = no caching)
    const size_t kMaxCached = getenv("PG_PINNED_CACHE_MB") ? (size_t)std::max(0L, atol(getenv("PG_PINNED_CACHE_MB"))) << 20 : (size_t)6 << 30;
    void *get(size_t bytes, size_t *got)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = free_blocks.size();
            for (size_t i = 0; i < free_blocks.size(); i++)
                if (free_blocks[i].second >= bytes && free_blocks[i].second <= 2 * bytes + 4096 &&
                    (best == free_blocks.size() || free_blocks[i].second < free_blocks[best].second))
                    best = i;
            if (best != free_blocks.size()) {
                void *p = free_blocks[best].first;
                *got = free_blocks[best].second;
                cached_bytes -= *got;
                free_blocks.erase(free_blocks.begin() + best);
                return p;
            }
        }
        void *p = nullptr;
        const size_t want = (bytes + 4095) & ~(size_t)4095;
        // portable: a cached block may be reused by a context on another GPU
        if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) return nullptr;
        *got = want;
        return p;
    }
    void put(void *p, size_t bytes)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (cached_bytes + bytes <= kMaxCached) {
                free_blocks.push_back(std::make_pair(p, bytes));
                cached_bytes += bytes;
                return;
            }
        }
        (void)hipHostFree(p);
    }
    // releases every cached block (when the last context of the process goes away)
    void trim()
    {
        std::vector<std::pair<void *, size_t>> gone;
        {
            std::lock_guard<std::mutex> lk(mu);
            gone.swap(free_blocks);
            cached_bytes = 0;
        }
        for (auto &b : gone) (void)hipHostFree(b.first);
    }
};
PinnedCache g_pinned;
std::mutex g_ctx_mu;
int g_live_ctx = 0;

template <typename T>
struct HostBuf {
    T *p = nullptr;
    size_t n = 0, cap_bytes = 0;
    bool view = false;                // p points into another HostBuf's block (pg_result::block): nothing to give back
    HostBuf() {}
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    ~HostBuf() { release(); }
    void release()
    {
        if (p && !view) g_pinned.put(p, cap_bytes);
        p = nullptr;
        n = cap_bytes = 0;
        view = false;
    }
    // count elements at `at` inside somebody else's pinned block (room for `room` elements: resize may shrink and regrow within it)
    void set_view(void *at, size_t count, size_t room)
    {
        release();
        p = (T *)at;
        n = count;
        cap_bytes = room * sizeof(T);
        view = true;
    }
    // contents are NOT initialised (they are about to be overwritten by a device-to-host copy)
    bool resize(size_t count)
    {
        if (count * sizeof(T) > cap_bytes) {
            release();
            p = (T *)g_pinned.get(std::max<size_t>(count * sizeof(T), 64), &cap_bytes);
            if (!p) return false;
        }
        n = count;
        return true;
    }
    bool assign(size_t count, T v)
    {
        if (!resize(count)) return false;
        for (size_t i = 0; i < count; i++) p[i] = v;
        return true;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    void swap(HostBuf &o)
    {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(cap_bytes, o.cap_bytes);
        std::swap(view, o.view);
    }
};

// Grow-only device arena of a ctx: the buffers of a host-path batch (pg_search_batch & co) are carved out of
// it instead of ~25 hipMalloc / hipFree per call.
struct DevArena {
    char *base = nullptr;
    size_t cap = 0, used = 0;
    void *take(size_t bytes)
    {
        const size_t at = (used + 255) & ~(size_t)255;
        if (at + bytes > cap) return nullptr;
        used = at + bytes;
        return base + at;
    }
};

}  // namespace

// The kernel keeps window bounds, anchor positions and candidate positions in signed 32-bit integers
// (pg_window, anchor_pos, `center`, `p`): a padded chromosome must stay below 2^31 with headroom for the
// guard/overhang arithmetic around a window end.
static const uint64_t PG_MAX_CHR_PADDED = 0x7fffffffull - 65536ull;

struct pg_ctx {
    pg_params prm{};
    uint32_t mm[512]{};
    uint16_t thr[512]{};
    uint16_t *d_thr = nullptr;
    uint32_t *d_mm = nullptr;
    PgLenRec *d_len_tab = nullptr;     // [512] pg_len_rec of every read length under this context's parameters (the pack's per-length fields)
    hipStream_t stream = nullptr, copy_stream = nullptr;   // kernels / host-to-device input copies
    hipStream_t dl_stream = nullptr;                       // device-to-host result copies of the chunked host path
    hipStream_t stream2 = nullptr;                         // second kernel stream of the chunked host path (odd chunks)
    std::vector<hipEvent_t> events;                        // reused by the chunked host path (no timing)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // reference
    std::vector<std::string> names;
    std::vector<uint64_t> comp_size;
    std::vector<uint64_t> word_off;   // index of the word holding AbsLoc 0
    std::vector<uint32_t> h_lo, h_hi, h_nn;
    uint32_t *d_lo = nullptr, *d_hi = nullptr, *d_nn = nullptr;
    uint64_t *d_word_off = nullptr;
    uint32_t *d_chr_size = nullptr;
    DevArena arena;                    // device memory of the host-path batches, reused across calls
    // stats of the last search
    double last_ms = 0.0;
    uint64_t last_runs = 0;
    std::string err;
    bool counted = false;              // in g_live_ctx (the pinned-memory cache is trimmed with the last context)
    uint32_t ref_epoch = 0;            // counts reference (re)loads: a device batch's records hold chromosome offsets and sizes
    bool kargs_checked = false;        // the kernels' view of the kernarg segment was checked on this device (pg_debug_kargs_check)
    bool last_in_place = false;        // the last launch asked to pack did so inside the search kernel (pg_debug_last_pack_in_place)
};

struct pg_device_batch {
    uint32_t n = 0;
    uint32_t max_len = 0, levels = 0;
    int32_t max_isz = 0;
    int64_t max_bd_window = 0;         // largest BreakDancer window attached (positions)
    uint64_t max_bd_cluster = 0;       // most windows attached to one read
    uint8_t *seq = nullptr;
    uint64_t *seq_off = nullptr;
    uint8_t *strand = nullptr;
    int32_t *pos = nullptr;
    int16_t *isz = nullptr;
    int32_t *chr = nullptr;
    uint8_t *rc_flag = nullptr;
    uint32_t *close_last = nullptr;
    uint16_t *close_max = nullptr;
    uint64_t *bd_off = nullptr;
    pg_window *bd = nullptr;
    uint32_t *close_off = nullptr, *close_cnt = nullptr, *far_off = nullptr, *far_cnt = nullptr;
    uint32_t *alg = nullptr;
    PgInRec *in_rec = nullptr;         // packed per-read records the kernel reads / writes (pg_device.h)
    uint32_t ref_epoch = 0;            // pg_ctx::ref_epoch when the records were packed (a reload of the reference: pack again)
    uint64_t *planes = nullptr;        // the reads as bit planes (PgDevBatch::planes)
    PgOutRec *out_rec = nullptr;
    bool unpacked = true;              // the SoA output arrays reflect out_rec
    bool in_arena = false;             // buffers belong to the ctx arena (not freed one by one)
    bool pool_owned = false;           // ... except a run pool that had to be regrown
    pg_run *pool = nullptr;
    uint32_t pool_shard_cap = 0;       // runs per shard (PG_POOL_SHARDS shards)
    uint32_t *pool_used = nullptr;     // [PG_POOL_SHARDS * 16]
    unsigned long long *run_tot = nullptr;   // running totals of the chunked delivery (zeroed with the outputs)
    uint64_t runs_used = 0;            // total runs of the last search
    int modes_done = 0;
    // reads with a character outside ACGTN, listed by the pack kernel for the exact kernel (pg_search_exact_kernel)
    uint32_t *exact_list = nullptr;    // [n]
    uint32_t *exact_count = nullptr;   // device counter (zeroed with the outputs / before a pack of the whole batch)
    PgSoaIn *d_soa = nullptr;          // the pack's view of the inputs, in device memory, for launches that pack in place (PgDevBatch::soa)
    PgSoaIn h_soa;                     // ... and what was copied there last (the source of an asynchronous copy)
    bool soa_uploaded = false;
    long long exact_n = -1;            // host copy of the counter; -1: not read back (the exact kernel is launched regardless)
};

struct pg_result {
    uint32_t n = 0;
    // one-chunk results of the host path arrive in ONE device-to-host copy: this block; the arrays below are views into it.
    // Declared BEFORE them: members are destroyed in reverse order of declaration, so the views go first and the block is handed
    // back to the pinned cache (where another thread may take it at once) only after nothing points into it any more.
    HostBuf<uint8_t> block;
    HostBuf<uint64_t> close_off, far_off;
    HostBuf<pg_run> close_runs, far_runs;
    HostBuf<uint8_t> rc_flag;
    HostBuf<uint32_t> close_last;
    HostBuf<uint16_t> close_max;
    HostBuf<uint32_t> csr32[2];        // staging of the device-built 32-bit offsets
};

namespace {

// One ctx drives one device; a process may hold several (one host thread each): every entry point selects
// its ctx's device for the calling thread first.
void use_device(const pg_ctx *ctx);

int fail(pg_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

#define HIP_TRY(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? PG_E_NOMEM : PG_E_DEVICE,           \
                        std::string(#call) + ": " + hipGetErrorString(e_));                  \
    } while (0)

template <typename T>
int dev_alloc(pg_ctx *ctx, T **p, size_t n)
{
    *p = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return PG_OK;
}

template <typename T>
int dev_upload(pg_ctx *ctx, T **p, const T *src, size_t n)
{
    int rc = dev_alloc(ctx, p, n);
    if (rc) return rc;
    if (n) HIP_TRY(ctx, hipMemcpy(*p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PG_OK;
}

// probOfReadWithTheseErrors / createProbTable, src/pindel.cpp:781-819
double prob_of_read_with_these_errors(unsigned length, unsigned n_err, double rate)
{
    double chance_correct = 1.0 - rate;
    unsigned n_correct = length - n_err;
    double matched = pow(chance_correct, (double)n_correct);
    double mismatched = 1.0;
    for (unsigned i = 0; i < n_err; i++) mismatched *= (((length - i) * rate) / (n_err - i));
    return matched * mismatched;
}

void make_tables(pg_ctx *ctx)
{
    const double rate = 0.001 + ctx->prm.seq_error_rate;   // pindel.cpp:856
    for (unsigned length = 0; length < 500; length++) {
        double total = 0.0;
        ctx->mm[length] = 0;
        for (unsigned n_err = 0; n_err <= length; n_err++) {
            total += prob_of_read_with_these_errors(length, n_err, rate);
            if (total > ctx->prm.sensitivity) {
                ctx->mm[length] = n_err + 1;
                break;
            }
        }
    }
    ctx->mm[0] = ctx->mm[1] = ctx->mm[2] = ctx->mm[3] = 0;
    for (unsigned length = 500; length < 512; length++) ctx->mm[length] = ctx->mm[499];
    // CheckMismatches: (float)NumMismatches >= (float)(len * MaximumAllowedMismatchRate),
    // src/searcher.cpp:366,383 -> smallest integer that passes, per read length
    for (unsigned length = 0; length < 512; length++) {
        float max_allowed = (float)((double)(size_t)length * ctx->prm.max_allowed_mismatch_rate);
        unsigned n = 0;
        while (!((float)n >= max_allowed) && n < 65535) n++;
        ctx->thr[length] = (uint16_t)n;
    }
}

void free_reference(pg_ctx *ctx)
{
    ctx->ref_epoch++;
    if (ctx->d_lo) (void)hipFree(ctx->d_lo);
    if (ctx->d_hi) (void)hipFree(ctx->d_hi);
    if (ctx->d_nn) (void)hipFree(ctx->d_nn);
    if (ctx->d_word_off) (void)hipFree(ctx->d_word_off);
    if (ctx->d_chr_size) (void)hipFree(ctx->d_chr_size);
    ctx->d_lo = ctx->d_hi = ctx->d_nn = nullptr;
    ctx->d_word_off = nullptr;
    ctx->d_chr_size = nullptr;
    ctx->names.clear();
    ctx->comp_size.clear();
    ctx->word_off.clear();
    ctx->h_lo.clear();
    ctx->h_hi.clear();
    ctx->h_nn.clear();
}

void use_device(const pg_ctx *ctx)
{
    if (ctx) (void)hipSetDevice(ctx->prm.device);
}

PgDevRef dev_ref(const pg_ctx *ctx)
{
    PgDevRef r;
    r.lo = ctx->d_lo;
    r.hi = ctx->d_hi;
    r.nn = ctx->d_nn;
    r.chr_word_off = ctx->d_word_off;
    r.chr_size = ctx->d_chr_size;
    r.n_chr = (int32_t)ctx->names.size();
    return r;
}

PgDevParams dev_params(const pg_ctx *ctx)
{
    PgDevParams p;
    p.max_range_index = ctx->prm.max_range_index;
    p.add_mm = ctx->prm.additional_mismatch;
    p.min_perfect = ctx->prm.min_perfect_match_around_bp;
    p.min_close = ctx->prm.min_close;
    p.spacer = ctx->prm.spacer;
    // breakpoints of the (monotone, checked in pg_create) g_maxMismatch table
    for (int k = 0; k < (int)PG_MM_BREAKS; k++) {
        p.mm_bp[k] = 0xffffffffu;
        for (unsigned L = 0; L < 500; L++)
            if (ctx->mm[L] >= (unsigned)(k + 1)) {
                p.mm_bp[k] = L;
                break;
            }
    }
    return p;
}

void free_batch_buffers(pg_device_batch *b)
{
    if (b->in_arena) {
        if (b->pool_owned && b->pool) (void)hipFree(b->pool);
        if (b->bd_off) (void)hipFree(b->bd_off);          // window clusters are always allocated on their own
        if (b->bd) (void)hipFree(b->bd);
        b->pool = nullptr;
        b->bd_off = nullptr;
        b->bd = nullptr;
        return;
    }
    void *ptrs[] = { b->planes, b->seq, b->seq_off, b->strand, b->pos, b->isz, b->chr, b->rc_flag,
                     b->close_last, b->close_max, b->bd_off, b->bd, b->close_off, b->close_cnt,
                     b->far_off, b->far_cnt, b->alg, b->pool, b->pool_used, b->in_rec, b->out_rec, b->run_tot,
                     b->exact_list, b->exact_count, b->d_soa };
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
}

// fn(lo, hi) over [0, n) in contiguous ranges on a few host threads (one for small n).  The threads are a process-wide
// pool that sleeps between calls: spawning sixteen std::threads per pass cost more than the pass (0.25 ms of the 1.5 ms a
// 4 M-read pg_search_batch spent before its first copy).  One job at a time; a second caller (another context's host
// thread in pg_search_batch_multi) that finds the pool busy runs its ranges on threads of its own as before.
//   * re-entry: a range function that calls host_ranges() again on a pool thread (or on the caller's) finds t_in_pool set and
//     runs its ranges inline (try_lock on a mutex the thread already owns would be undefined behaviour);
//   * fork(): the child has the pool object but none of its threads; the pool lives behind a pointer and a pthread_atfork
//     child handler replaces it with a fresh one (the old object is leaked on purpose: joining threads that do not exist in
//     this process, or destroying mutexes another thread held at the fork, is not defined);
//   * a range function must not throw across threads: an exception on a worker is caught, the job completes, and the first
//     one is rethrown on the calling thread.
thread_local bool t_in_pool = false;
class HostPool {
public:
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    bool try_run(size_t n, unsigned nt, const std::function<void(size_t, size_t)> &fn)
    {
        if (t_in_pool) return false;                            // re-entry from a range function
        std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
        if (!job.owns_lock()) return false;
        struct InPool { InPool() { t_in_pool = true; } ~InPool() { t_in_pool = false; } } in_pool;
        {
            std::lock_guard<std::mutex> lk(mu_);
            while (th_.size() + 1 < nt) {
                const unsigned id = (unsigned)th_.size() + 1;
                th_.emplace_back([this, id] { worker(id); });
            }
            fn_ = &fn;
            n_ = n;
            nt_ = nt;
            remaining_ = nt - 1;
            error_ = nullptr;
            gen_++;
        }
        cv_.notify_all();
        std::exception_ptr mine;
        try {
            fn((size_t)0, n / nt);                              // the caller takes the first range
        } catch (...) {
            mine = std::current_exception();
        }
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return remaining_ == 0; });
        fn_ = nullptr;
        std::exception_ptr err = mine ? mine : error_;
        error_ = nullptr;
        lk.unlock();
        if (err) std::rethrow_exception(err);
        return true;
    }
private:
    void worker(unsigned id)
    {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(size_t, size_t)> *fn;
            size_t n;
            unsigned nt;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (gen_ != seen && id < nt_); });
                if (stop_) return;
                seen = gen_;
                fn = fn_;
                n = n_;
                nt = nt_;
            }
            std::exception_ptr err;
            t_in_pool = true;
            try {
                (*fn)(n * id / nt, n * (id + 1) / nt);
            } catch (...) {
                err = std::current_exception();
            }
            std::lock_guard<std::mutex> lk(mu_);
            if (err && !error_) error_ = err;
            if (--remaining_ == 0) done_.notify_one();
        }
    }
    std::mutex job_mu_, mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    const std::function<void(size_t, size_t)> *fn_ = nullptr;
    size_t n_ = 0;
    unsigned nt_ = 0, remaining_ = 0, gen_ = 0;
    bool stop_ = false;
    std::exception_ptr error_;
};
HostPool *g_host_pool = nullptr;
std::once_flag g_host_pool_once;
HostPool &host_pool()
{
    std::call_once(g_host_pool_once, [] {
        g_host_pool = new HostPool();
        (void)pthread_atfork(nullptr, nullptr, [] { g_host_pool = new HostPool(); t_in_pool = false; });
        atexit([] { HostPool *p = g_host_pool; g_host_pool = nullptr; delete p; });
    });
    return *g_host_pool;
}

template <class Fn>
void host_ranges(size_t n, Fn fn)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::min<size_t>(std::min(hw, 16u), n >> 16);
    if (nt <= 1) {
        fn((size_t)0, n);
        return;
    }
    if (t_in_pool) {                                            // re-entry from a range function: inline, no nt x nt fan-out
        fn((size_t)0, n);
        return;
    }
    if (host_pool().try_run(n, nt, std::function<void(size_t, size_t)>(fn))) return;
    // the pool is busy with another caller's job: threads of this call's own, exceptions carried back like the pool's
    std::vector<std::thread> th;
    std::mutex err_mu;
    std::exception_ptr err;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            try {
                fn(n * t / nt, n * (t + 1) / nt);
            } catch (...) {
                std::lock_guard<std::mutex> lk(err_mu);
                if (!err) err = std::current_exception();
            }
        });
    for (std::thread &x : th) x.join();
    if (err) std::rethrow_exception(err);
}

// max_isz (nullable): largest insert size of the batch
// off_out (nullable, n + 1 entries): the read offsets rebased to 0, written in the same pass
int validate_and_measure(pg_ctx *ctx, const pg_read_batch *reads, uint32_t *max_len, uint32_t *levels, int32_t *max_isz = nullptr,
                         uint64_t *off_out = nullptr)
{
    if (!reads || (reads->n_reads && (!reads->seq_off || !reads->anchor_strand || !reads->anchor_pos ||
                                      !reads->insert_size || !reads->chr_id)))
        return fail(ctx, PG_E_INVALID, "null array in pg_read_batch");
    if (reads->n_reads && !reads->seq && reads->seq_off[reads->n_reads] > reads->seq_off[0])
        return fail(ctx, PG_E_INVALID, "null seq in pg_read_batch");
    if (ctx->names.empty()) return fail(ctx, PG_E_NO_REFERENCE, "no reference loaded");
    const int n_chr = (int)ctx->names.size();
    // per range: first problem found (0 = none; the lowest code wins, as in a sequential scan the first one would), longest read
    struct Part { int code = 0; uint32_t ml = 1; int32_t isz = 0; };
    std::mutex mu;
    Part all;
    size_t first_bad = (size_t)-1;
    const uint64_t base0 = reads->n_reads ? reads->seq_off[0] : 0;
    if (off_out) off_out[reads->n_reads] = reads->n_reads ? reads->seq_off[reads->n_reads] - base0 : 0;
    host_ranges(reads->n_reads, [&](size_t lo, size_t hi) {
        Part p;
        size_t bad = (size_t)-1;
        for (size_t i = lo; i < hi; i++) {
            int code = 0;
            if (off_out) off_out[i] = reads->seq_off[i] - base0;
            if (reads->seq_off[i + 1] < reads->seq_off[i]) code = 1;
            const uint64_t len = reads->seq_off[i + 1] - reads->seq_off[i];
            if (!code && len > PG_MAX_READ_LEN) code = 2;
            const int c = reads->chr_id[i];
            if (!code && (c < 0 || c >= n_chr)) code = 3;
            if (!code) {
                // the close-end windows (pindel.cpp:2272-2274, 2301-2302) must lie inside the padded string
                const long long apos = (long long)reads->anchor_pos[i] + ctx->prm.spacer;
                const long long isz = reads->insert_size[i];
                const long long wlo = apos - 2 * std::max<long long>(isz, 0) - 2, whi = apos + 2 * std::max<long long>(isz, 0) + 2;
                if (wlo < 0 || whi > (long long)ctx->comp_size[c]) code = 4;
            }
            if (code) {
                p.code = code;
                bad = i;
                break;
            }
            p.ml = std::max<uint32_t>(p.ml, (uint32_t)len);
            p.isz = std::max<int32_t>(p.isz, reads->insert_size[i]);
        }
        std::lock_guard<std::mutex> lk(mu);
        all.ml = std::max(all.ml, p.ml);
        all.isz = std::max(all.isz, p.isz);
        if (p.code && bad < first_bad) {
            first_bad = bad;
            all.code = p.code;
        }
    });
    switch (all.code) {
    case 1: return fail(ctx, PG_E_INVALID, "seq_off not monotone");
    case 2: return fail(ctx, PG_E_READ_TOO_LONG, "read longer than 499 bases");
    case 3: return fail(ctx, PG_E_INVALID, "chr_id out of range");
    case 4: return fail(ctx, PG_E_INVALID, "anchor position/insert size reach outside the padded chromosome");
    default: break;
    }
    const uint32_t ml = all.ml;
    *max_len = ml;
    if (max_isz) *max_isz = all.isz;
    uint32_t lv = ctx->mm[ml] + (uint32_t)ctx->prm.additional_mismatch + 1;
    for (uint32_t l = 0; l <= ml; l++)
        lv = std::max<uint32_t>(lv, ctx->mm[l] + (uint32_t)ctx->prm.additional_mismatch + 1);
    if (lv > PG_MAX_LEVELS) return fail(ctx, PG_E_UNSUPPORTED, "more than 32 mismatch levels");
    *levels = lv;
    return PG_OK;
}

// Runs per list the chunked delivery of search_host has room for (1.04 per read on average; a batch that needs more
// falls back to the whole-batch download).
static size_t deliver_cap(size_t n) { return env().tiny_delivery ? n / 2 + 8 : 2 * n + 4096; }   // (tests: force the fallback)

// Validates the batch and allocates its device buffers.  copy = true also copies the inputs
// (synchronously); otherwise the caller streams them in (search_host).  off = read offsets rebased to 0.
// use_arena: carve the buffers out of the ctx arena (host-path calls: the batch dies with the call).
PgSoaIn soa_in(const pg_ctx *ctx, const pg_device_batch *b);
int alloc_batch(pg_ctx *ctx, const pg_read_batch *reads, bool copy, HostBuf<uint64_t> &off, pg_device_batch **out,
                bool use_arena = false)
{
    uint32_t max_len = 0, levels = 0;
    int32_t max_isz = 0;
    // (room behind the offsets for the four small per-read arrays in the arena's layout: search_host sends a one-chunk batch's
    // small inputs with ONE copy from this pinned block)
    const size_t n_in = reads ? reads->n_reads : 0;
    if (!off.resize(n_in + 1 + (use_arena ? (n_in * 11 + 6 * 512) / 8 + 8 : 0))) return fail(ctx, PG_E_NOMEM, "host memory for the read offsets");
    int rc = validate_and_measure(ctx, reads, &max_len, &levels, &max_isz, off.data());
    if (rc) return rc;
    pg_device_batch *b = new pg_device_batch();
    b->n = reads->n_reads;
    // the batch is bound to the reference it was VALIDATED against (anchor windows, chromosome ids): stamped here and nowhere else --
    // a repack or a new set of windows after a reload of the reference is refused, not re-stamped (stale_batch)
    b->ref_epoch = ctx->ref_epoch;
    b->max_len = max_len;
    b->levels = levels;
    b->max_isz = max_isz;
    const size_t n = b->n;
    const uint64_t base0 = n ? reads->seq_off[0] : 0;
    const uint64_t nseq = n ? reads->seq_off[n] - base0 : 0;
    // every read reserves PG_RESERVE slots (one atomic per claim of reads); lists longer than their share allocate more
    b->pool_shard_cap = (uint32_t)std::min<uint64_t>(((PG_RESERVE + 2ull) * n) / PG_POOL_SHARDS + 512ull, 0x7fffffffull / PG_POOL_SHARDS);
    if (env().tiny_pool) b->pool_shard_cap = 1;      // tests: force the overflow/regrow path
    const size_t n1 = std::max<size_t>(n, 1);
    // (pointer, bytes) of every buffer; the zeroed block (outputs) is contiguous in the arena case
    struct Item { void **p; size_t bytes; };
    const Item items[] = {
        { (void **)&b->seq, (size_t)nseq + 16 }, { (void **)&b->seq_off, (n + 1) * 8 },
        { (void **)&b->strand, n1 }, { (void **)&b->pos, n1 * 4 }, { (void **)&b->isz, n1 * 2 }, { (void **)&b->chr, n1 * 4 },
        // INVARIANT: every allocation of in_rec carries PG_IN_PAD records behind the last read -- search_read touches `rp + 1` with a
        // scalar load whose value is discarded (the next read's record into the scalar cache); this is the only place in_rec is
        // allocated, and a launch over a sub-range [lo, lo + cnt) of the batch touches at most record lo + cnt, which exists.
        { (void **)&b->in_rec, (n1 + PG_IN_PAD) * sizeof(PgInRec) },
        { (void **)&b->planes, n1 * 64 * pg_plane_blocks(max_len) },
        { (void **)&b->exact_list, n1 * 4 },
        { (void **)&b->d_soa, sizeof(PgSoaIn) },
        { (void **)&b->pool, (size_t)b->pool_shard_cap * PG_POOL_SHARDS * sizeof(pg_run) },
        // ---- zero-initialised from here
        { (void **)&b->rc_flag, n1 }, { (void **)&b->close_last, n1 * 4 }, { (void **)&b->close_max, n1 * 2 },
        { (void **)&b->close_off, n1 * 4 }, { (void **)&b->close_cnt, (n + 1) * 4 },   // + 1: the CSR scan runs over n + 1 counts
        { (void **)&b->far_off, n1 * 4 }, { (void **)&b->far_cnt, (n + 1) * 4 }, { (void **)&b->alg, n1 * 4 },
        { (void **)&b->out_rec, n1 * sizeof(PgOutRec) },
        // run-pool cursors, then per set of read counters (two: launches on the two kernel streams overlap) the counters and the
        // cycle accumulators of a -DPG_TIMING diagnostics build, which the kernel finds right behind its counters
        { (void **)&b->pool_used, (PG_POOL_SHARDS * 16 + 2 * (PG_WORK_CTRS * 16 + PG_DIAG_WORDS * 2)) * 4 },   // + a second set of read counters  // run-pool cursors + the launch's read counters
        { (void **)&b->run_tot, 64 },
        { (void **)&b->exact_count, 64 },
    };
    const size_t n_items = sizeof items / sizeof items[0], first_zero = 11;
    auto drop = [&](int code) {
        free_batch_buffers(b);
        delete b;
        return code;
    };
    if (use_arena) {
        size_t need = 4096;
        for (const Item &it : items) need += ((it.bytes + 255) & ~(size_t)255) + 256;
        // room for the download's temporaries too: CSR offsets, gathered runs, scan scratch
        need += 2 * (((n + 1) * 4 + 511) & ~(size_t)255) + (size_t)b->pool_shard_cap * PG_POOL_SHARDS * sizeof(pg_run) +
                pg_scan_tmp_bytes((uint32_t)n) + 4096;
        // ... and for the chunk-by-chunk delivery of search_host: gathered runs (2 lists), 64-bit offsets (2 lists), scratch
        need += 2 * (deliver_cap(n) * sizeof(pg_run) + 512) + 2 * ((n + 1) * 8 + 512) + PG_DELIVER_CHUNK * 8 + 4096 * 8 +
                (n / std::min<size_t>(PG_HOST_CHUNK / 4, env().host_chunk ? env().host_chunk : PG_HOST_CHUNK) + 16) * 64 +   // 64 B of info per chunk
                8192 + 8 * n + 4096;                                           // (+ the summaries of a one-block delivery)
        if (need > ctx->arena.cap) {
            if (ctx->arena.base) (void)hipFree(ctx->arena.base);
            ctx->arena.base = nullptr;
            ctx->arena.cap = 0;
            const size_t want = need + need / 4;
            hipError_t e = hipMalloc((void **)&ctx->arena.base, want);
            if (e != hipSuccess) {
                delete b;
                return fail(ctx, PG_E_NOMEM, std::string("device arena: ") + hipGetErrorString(e));
            }
            ctx->arena.cap = want;
        }
        ctx->arena.used = 0;
        b->in_arena = true;
        for (const Item &it : items) *it.p = ctx->arena.take(it.bytes);
        char *z0 = (char *)*items[first_zero].p;
        char *z1 = (char *)*items[n_items - 1].p + items[n_items - 1].bytes;
        hipError_t e = hipMemsetAsync(z0, 0, (size_t)(z1 - z0), ctx->stream);
        if (e != hipSuccess) return drop(fail(ctx, PG_E_DEVICE, std::string("clearing the batch outputs: ") + hipGetErrorString(e)));
    } else {
        for (size_t k = 0; k < n_items; k++) {
            hipError_t e = hipMalloc(items[k].p, items[k].bytes);
            // the device-side CSR scan / gather trusts the counts: a failed memset must not go unnoticed
            if (e == hipSuccess && k >= first_zero) e = hipMemset(*items[k].p, 0, items[k].bytes);
            if (e != hipSuccess)
                return drop(fail(ctx, e == hipErrorOutOfMemory ? PG_E_NOMEM : PG_E_DEVICE,
                                 std::string("batch buffers: ") + hipGetErrorString(e)));
        }
    }
    if (copy && n) {
        hipError_t e = hipSuccess;
        if (nseq) e = hipMemcpy(b->seq, reads->seq + base0, (size_t)nseq, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->seq_off, off.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->strand, reads->anchor_strand, n, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->pos, reads->anchor_pos, n * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->isz, reads->insert_size, n * sizeof(int16_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->chr, reads->chr_id, n * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) return drop(fail(ctx, PG_E_DEVICE, std::string("input upload: ") + hipGetErrorString(e)));
    }
    // the pack's view of the inputs for launches that pack in place (launch_range replaces it when the windows or the parameters
    // change); on the ctx stream, which every launch of the batch is ordered behind
    b->h_soa = soa_in(ctx, b);
    {
        hipError_t e = hipMemcpyAsync(b->d_soa, &b->h_soa, sizeof b->h_soa, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return drop(fail(ctx, PG_E_DEVICE, std::string("input upload: ") + hipGetErrorString(e)));
    }
    b->soa_uploaded = true;
    *out = b;
    return PG_OK;
}

PgSoaIn soa_in(const pg_ctx *ctx, const pg_device_batch *b)
{
    PgSoaIn a;
    std::memset(&a, 0, sizeof a);      // (compared bytewise by launch_range)
    a.seq = b->seq;
    a.planes = b->planes;
    a.plane_blocks = pg_plane_blocks(b->max_len);
    a.seq_off = b->seq_off;
    a.strand = b->strand;
    a.pos = b->pos;
    a.isz = b->isz;
    a.chr = b->chr;
    a.bd_off = b->bd_off;
    a.exact_list = b->exact_list;
    a.exact_count = b->exact_count;
    a.len_tab = ctx->d_len_tab;
    a.chr_word_off = ctx->d_word_off;
    a.chr_size = ctx->d_chr_size;
    a.spacer = ctx->prm.spacer;
    a.add_mm = ctx->prm.additional_mismatch;
    a.min_close = ctx->prm.min_close;
    return a;
}

PgSoaOut soa_out(const pg_device_batch *b)
{
    PgSoaOut a;
    a.rc_flag = b->rc_flag;
    a.close_last = b->close_last;
    a.close_max = b->close_max;
    a.close_off = b->close_off;
    a.close_cnt = b->close_cnt;
    a.far_off = b->far_off;
    a.far_cnt = b->far_cnt;
    a.alg = b->alg;
    a.cand = nullptr;
    return a;
}

// A device batch whose reference has been replaced since it was validated: its chromosome ids, anchor windows and (once packed)
// word offsets belong to the old reference.
int stale_batch(pg_ctx *ctx, const pg_device_batch *b)
{
    if (b->ref_epoch != ctx->ref_epoch && b->n)
        return fail(ctx, PG_E_INVALID, "the reference was (re)loaded after this batch was uploaded: upload the batch again");
    return PG_OK;
}

// The length of the exact kernel's list, once the batch's pack has finished (synchronous entries only: a device-resident batch is
// searched many times, and every search would otherwise carry a launch that finds nothing to do).
int read_exact_count(pg_ctx *ctx, pg_device_batch *b)
{
    uint32_t n = 0;
    if (b->n) {
        HIP_TRY(ctx, hipMemcpyAsync(&n, b->exact_count, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    b->exact_n = n;
    return PG_OK;
}

// Builds the packed records of reads [lo, lo + cnt) from the SoA inputs (on the ctx stream).
int pack_reads(pg_ctx *ctx, pg_device_batch *b, uint32_t lo, uint32_t cnt, hipStream_t st = nullptr)
{
    const PgSoaIn a = soa_in(ctx, b);
    if (lo == 0 && cnt == b->n && b->n) {       // the whole batch (again): the exact kernel's list starts empty
        HIP_TRY(ctx, hipMemsetAsync(b->exact_count, 0, sizeof(uint32_t), st ? st : ctx->stream));
        b->exact_n = -1;
    }
    int rc = pg_pack_reads(&a, b->in_rec, lo, cnt, st ? st : ctx->stream);
    if (rc) return fail(ctx, PG_E_DEVICE, std::string("pack kernel: ") + hipGetErrorString((hipError_t)rc));
    return PG_OK;
}

// Scatters the kernel's output records into the SoA arrays the CSR scan / download read.
int unpack_results(pg_ctx *ctx, pg_device_batch *b)
{
    if (b->unpacked) return PG_OK;
    const PgSoaOut a = soa_out(b);
    int rc = pg_unpack_results(b->out_rec, &a, b->n, ctx->stream);
    if (rc) return fail(ctx, PG_E_DEVICE, std::string("unpack kernel: ") + hipGetErrorString((hipError_t)rc));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    b->unpacked = true;
    return PG_OK;
}

int upload_batch(pg_ctx *ctx, const pg_read_batch *reads, pg_device_batch **out)
{
    // (pinned, from the cache: as a malloc'd array the runtime locks its pages for the copy and the free() of 80 MB after a
    // 10 M-read call spent 13 ms in munmap + unpinning)
    HostBuf<uint64_t> off;
    int rc = alloc_batch(ctx, reads, true, off, out);
    if (rc) return rc;
    if ((rc = pack_reads(ctx, *out, 0, (*out)->n))) {
        free_batch_buffers(*out);
        delete *out;
        *out = nullptr;
    }
    return rc;
}

PgDevBatch dev_batch(const pg_device_batch *b)
{
    PgDevBatch d;
    d.n_reads = b->n;
    d.first_read = 0;
    d.in = b->in_rec;
    d.out = b->out_rec;
    d.seq = b->seq;
    d.planes = b->planes;
    d.plane_blocks = pg_plane_blocks(b->max_len);
    d.bd = b->bd_off ? b->bd : nullptr;
    d.pool = b->pool;
    d.pool_shard_cap = b->pool_shard_cap;
    d.pool_used = b->pool_used;
    d.work_ctr = b->pool_used + PG_POOL_SHARDS * 16;
    d.claim = PG_CLAIM_DEFAULT;        // (pg_launch_search sets it from the launch's grid)
    d.exact_list = b->exact_list;
    d.exact_count = b->exact_count;
    d.thr_tab = nullptr;               // (launch_range: the ctx's table)
    d.soa = nullptr;                   // (launch_range: a launch that packs in place)
    return d;
}

bool small_ids(const pg_ctx *ctx, const pg_device_batch *b)
{
    // 32-bit candidate ids when every window of this launch has <= 2^24 positions: ranges 128 * 4^x
    // (x <= 8), close windows 3 * InsertSize (a short), BreakDancer windows as attached
    return ctx->prm.max_range_index <= 8 && (long long)b->max_bd_window <= PG_SMALL_MAX_WINDOW &&
           b->max_bd_cluster <= PG_SMALL_MAX_CLUSTER && !env().force_wide_cells;
}

// Launches the search for reads [lo, lo + cnt) of the batch on the ctx stream.
// st / set: the stream to launch on and the set of read counters to use (launches that may overlap need their own)
// fresh: the set's counters are still zero from the batch's allocation (the first launch on each set)
// pack: the records of the range are (re)built from the batch's SoA inputs first -- by the search kernel itself where that is
// possible (pg_pack_in_place_ok: PgDevBatch::soa), by a pack launch in front of it otherwise
int launch_range(pg_ctx *ctx, pg_device_batch *b, int mode, uint32_t lo, uint32_t cnt, hipStream_t st = nullptr, int set = 0,
                 bool fresh = false, bool pack = false)
{
    PgDevRef ref = dev_ref(ctx);
    PgDevParams prm = dev_params(ctx);
    PgDevBatch d = dev_batch(b);
    d.first_read = lo;
    d.n_reads = cnt;
    if (!st) st = ctx->stream;
    // the persistent launch claims its reads from these counters (set 1 lies behind set 0 and the diagnostics words)
    size_t bytes = (PG_WORK_CTRS * 16 + PG_DIAG_WORDS * 2) * sizeof(uint32_t);
    if (set) {
        d.work_ctr += PG_WORK_CTRS * 16 + PG_DIAG_WORDS * 2;
        bytes = PG_WORK_CTRS * 16 * sizeof(uint32_t);
    }
    if (!ctx->kargs_checked) {
        // once per context: the kernels fetch their arguments from the kernarg segment at offsets of their own (KA() in
        // pg_kernels.hip); a one-wave kernel with the same parameter list compares them with the by-value arguments
        const int bad = pg_debug_kargs_check(&ref, &prm, &d, b->max_len, b->levels, d.work_ctr + PG_WORK_CTRS * 16, st);
        if (bad != 0) return fail(ctx, PG_E_DEVICE, "kernel arguments: the kernarg segment is not laid out as the kernels expect (" + std::to_string(bad) + ")");
        ctx->kargs_checked = true;
    }
    if (!fresh) HIP_TRY(ctx, hipMemsetAsync(d.work_ctr, 0, bytes, st));
    d.thr_tab = ctx->d_thr;
    if (pack && cnt) {
        if (pg_pack_in_place_ok(mode, b->max_len, small_ids(ctx, b) ? 1 : 0, cnt, d.plane_blocks)) {
            const PgSoaIn a = soa_in(ctx, b);
            if (std::memcmp(&a, &b->h_soa, sizeof a) != 0 || !b->soa_uploaded) {     // (the parameters or the windows changed)
                // (launches that overlap on the two kernel streams read the same copy: wait for them before it is replaced)
                if (b->soa_uploaded) HIP_TRY(ctx, hipDeviceSynchronize());
                b->h_soa = a;
                // (blocking: the chunks of the host path alternate between two kernel streams, and every one of them reads this copy)
                HIP_TRY(ctx, hipMemcpy(b->d_soa, &b->h_soa, sizeof a, hipMemcpyHostToDevice));
                b->soa_uploaded = true;
            }
            if (lo == 0 && cnt == b->n) {                 // the whole batch (again): the exact kernel's list starts empty
                HIP_TRY(ctx, hipMemsetAsync(b->exact_count, 0, sizeof(uint32_t), st));
                b->exact_n = -1;
            }
            d.soa = b->d_soa;
        } else if (int rc = pack_reads(ctx, b, lo, cnt, st))
            return rc;
        ctx->last_in_place = d.soa != nullptr;
    }
    int lrc = pg_launch_search(&ref, &prm, &d, mode, b->max_len, b->levels, small_ids(ctx, b) ? 1 : 0, st);
    if (lrc != 0) return fail(ctx, PG_E_DEVICE, std::string("kernel launch: ") + hipGetErrorString((hipError_t)lrc));
    // ... then, behind it on the same stream, the reads of this range that hold a character outside ACGTN, with the reference's
    // read-shortening semantics (setUnmatchedSeq, pindel.cpp:142-169, 2545): a 64-workgroup launch that finds its list empty for
    // every batch a sequencer produced -- skipped when the host has read the list's length and it is zero
    if (b->exact_n != 0) {
        lrc = pg_launch_search_exact(&ref, &prm, &d, mode, b->max_len, b->levels, st);
        if (lrc != 0) return fail(ctx, PG_E_DEVICE, std::string("exact kernel launch: ") + hipGetErrorString((hipError_t)lrc));
    }
    return PG_OK;
}

// After the launches of a search: total runs, and whether a pool shard overflowed (worst > capacity).
int read_cursors(pg_ctx *ctx, pg_device_batch *b, uint32_t *worst, uint64_t *total)
{
    std::vector<uint32_t> cursors(PG_POOL_SHARDS * 16);
    HIP_TRY(ctx, hipMemcpy(cursors.data(), b->pool_used, cursors.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *worst = 0;
    *total = 0;
    for (uint32_t s = 0; s < PG_POOL_SHARDS; s++) {
        *worst = std::max(*worst, cursors[s * 16]);
        *total += cursors[s * 16];
    }
    return PG_OK;
}

// Runs the kernel(s) of `mode`; if a pool shard overflowed, the pool is regrown and the launch
// repeated (still entirely on the GPU).
int run_search(pg_ctx *ctx, pg_device_batch *b, int mode, bool pack = false)
{
    if (ctx->names.empty()) return fail(ctx, PG_E_NO_REFERENCE, "no reference loaded");
    // (a batch is validated and its records are packed against the reference loaded at upload time: chromosome offsets and sizes,
    // window bounds; a reference loaded later makes them stale)
    if (int rc = stale_batch(ctx, b)) return rc;
    for (int attempt = 0; attempt < 8; attempt++) {
        HIP_TRY(ctx, hipMemsetAsync(b->pool_used, 0, PG_POOL_SHARDS * 16 * sizeof(uint32_t), ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        int rc = launch_range(ctx, b, mode, 0, b->n, nullptr, 0, false, pack);
        if (rc) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        uint32_t worst = 0;
        uint64_t total = 0;
        if ((rc = read_cursors(ctx, b, &worst, &total))) return rc;
        if (worst <= b->pool_shard_cap) {
            ctx->last_ms = ms;
            ctx->last_runs = total;
            b->runs_used = total;
            b->modes_done |= mode;
            b->unpacked = false;
            return PG_OK;
        }
        // overflow: grow the pool and redo the launch
        uint32_t ncap = (uint32_t)std::min<uint64_t>((uint64_t)worst + worst / 4 + 64, 0x7fffffffull / PG_POOL_SHARDS);
        pg_run *npool = nullptr;
        rc = dev_alloc(ctx, &npool, (size_t)ncap * PG_POOL_SHARDS);
        if (rc) return rc;
        if (!b->in_arena || b->pool_owned) (void)hipFree(b->pool);
        b->pool = npool;
        b->pool_owned = true;
        b->pool_shard_cap = ncap;
    }
    return fail(ctx, PG_E_DEVICE, "run pool kept overflowing");
}

// Results to the host: the runs are gathered into read order on the device (prefix sums of the per-read
// counts + one gather kernel per list), so only the compact CSR crosses PCIe; the host buffers are pinned.
int download(pg_ctx *ctx, pg_device_batch *b, pg_result *r)
{
    const size_t n = b->n;
    r->n = b->n;
    const bool has[2] = { (b->modes_done & PG_MODE_CLOSE) != 0, (b->modes_done & PG_MODE_FAR) != 0 };
    if (!r->close_off.resize(n + 1) || !r->far_off.resize(n + 1) || !r->rc_flag.resize(n) || !r->close_last.resize(n) ||
        !r->close_max.resize(n) || !r->csr32[0].resize(n + 1) || !r->csr32[1].resize(n + 1))
        return fail(ctx, PG_E_NOMEM, "pinned host memory for the result");
    r->close_runs.resize(0);
    r->far_runs.resize(0);
    if (!n) {
        r->close_off[0] = r->far_off[0] = 0;
        return PG_OK;
    }
    {
        int urc = unpack_results(ctx, b);
        if (urc) return urc;
    }
    uint32_t *csr[2] = { nullptr, nullptr };
    pg_run *outp[2] = { nullptr, nullptr };
    void *tmp = nullptr;
    std::vector<void *> owned;                            // hipMalloc'd temporaries (none in the arena case)
    auto dev_take = [&](size_t bytes) -> void * {
        if (b->in_arena) {
            void *p = ctx->arena.take(bytes);
            if (p) return p;
        }
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
        owned.push_back(p);
        return p;
    };
    auto cleanup = [&](int code) {
        for (void *p : owned) (void)hipFree(p);
        return code;
    };
#define TRY2(call)                                                                           \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return cleanup(fail(ctx, e_ == hipErrorOutOfMemory ? PG_E_NOMEM : PG_E_DEVICE,   \
                                std::string(#call) + ": " + hipGetErrorString(e_)));         \
    } while (0)
    const size_t arena_mark = ctx->arena.used;
    const size_t tmp_bytes = pg_scan_tmp_bytes((uint32_t)n);
    if (!(tmp = dev_take(tmp_bytes))) return cleanup(fail(ctx, PG_E_NOMEM, "scan scratch"));
    const uint32_t *cnts[2] = { b->close_cnt, b->far_cnt }, *offs[2] = { b->close_off, b->far_off };
    uint32_t totals[2] = { 0, 0 };
    for (int k = 0; k < 2; k++) {
        if (!has[k]) continue;
        if (!(csr[k] = (uint32_t *)dev_take((n + 1) * sizeof(uint32_t)))) return cleanup(fail(ctx, PG_E_NOMEM, "CSR offsets"));
        TRY2((hipError_t)pg_compact_runs(nullptr, nullptr, cnts[k], csr[k], nullptr, (uint32_t)n, tmp, tmp_bytes, 0, ctx->stream));
        TRY2(hipMemcpyAsync(r->csr32[k].data(), csr[k], (n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    // the per-read summaries travel while the scans run
    TRY2(hipMemcpyAsync(r->rc_flag.data(), b->rc_flag, n, hipMemcpyDeviceToHost, ctx->stream));
    TRY2(hipMemcpyAsync(r->close_last.data(), b->close_last, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TRY2(hipMemcpyAsync(r->close_max.data(), b->close_max, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    TRY2(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 2; k++) {
        HostBuf<pg_run> &runs = k ? r->far_runs : r->close_runs;
        HostBuf<uint64_t> &off64 = k ? r->far_off : r->close_off;
        if (!has[k]) {
            for (size_t i = 0; i <= n; i++) off64[i] = 0;
            continue;
        }
        totals[k] = r->csr32[k][n];
        if (!runs.resize(totals[k])) return cleanup(fail(ctx, PG_E_NOMEM, "pinned host memory for the runs"));
        if (totals[k]) {
            if (!(outp[k] = (pg_run *)dev_take((size_t)totals[k] * sizeof(pg_run)))) return cleanup(fail(ctx, PG_E_NOMEM, "gathered runs"));
            TRY2((hipError_t)pg_compact_runs(b->pool, offs[k], cnts[k], csr[k], outp[k], (uint32_t)n, nullptr, 0, 1, ctx->stream));
            TRY2(hipMemcpyAsync(runs.data(), outp[k], (size_t)totals[k] * sizeof(pg_run), hipMemcpyDeviceToHost, ctx->stream));
        }
        const uint32_t *c32 = r->csr32[k].data();
        for (size_t i = 0; i <= n; i++) off64[i] = c32[i];       // widening while the runs are in flight
    }
    TRY2(hipStreamSynchronize(ctx->stream));
#undef TRY2
    if (b->in_arena) ctx->arena.used = arena_mark;
    return cleanup(PG_OK);
}

// Copies the packed planes and the chromosome tables of the ctx to the device.
int upload_reference(pg_ctx *ctx)
{
    std::vector<uint32_t> sizes(ctx->comp_size.size());
    for (size_t c = 0; c < sizes.size(); c++) sizes[c] = (uint32_t)ctx->comp_size[c];
    int rc;
    if ((rc = dev_upload(ctx, &ctx->d_lo, ctx->h_lo.data(), ctx->h_lo.size())) ||
        (rc = dev_upload(ctx, &ctx->d_hi, ctx->h_hi.data(), ctx->h_hi.size())) ||
        (rc = dev_upload(ctx, &ctx->d_nn, ctx->h_nn.data(), ctx->h_nn.size())) ||
        (rc = dev_upload(ctx, &ctx->d_word_off, ctx->word_off.data(), ctx->word_off.size())) ||
        (rc = dev_upload(ctx, &ctx->d_chr_size, sizes.data(), sizes.size()))) {
        free_reference(ctx);
        return rc;
    }
    return PG_OK;
}

}  // namespace

// ============================================================================== C ABI
extern "C" {

const PgEnvSwitches *pg_env_switches(void) { return &env(); }
void pg_debug_reload_env(void)
{
    (void)env();
    load_env();
}

void pg_default_params(pg_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->abi_version = PG_ABI_VERSION;
    p->device = 0;
    p->max_range_index = 2;
    p->additional_mismatch = 1;
    p->min_perfect_match_around_bp = 3;
    p->min_close = 8;
    p->max_allowed_mismatch_rate = 0.02;
    p->seq_error_rate = 0.01;
    p->sensitivity = 0.95;
    p->spacer = 100000;
}

int pg_create(const pg_params *p, pg_ctx **out)
{
    if (!p || !out) return PG_E_INVALID;
    *out = nullptr;
    if (p->abi_version != PG_ABI_VERSION) return PG_E_INVALID;
    pg_ctx *ctx = new pg_ctx();
    ctx->prm = *p;
    // pindel.cpp:921-930: -x capped at g_MAX_RANGE_INDEX (9), -a raised to 1
    if (ctx->prm.max_range_index > 9) ctx->prm.max_range_index = 9;
    if (ctx->prm.max_range_index < 0) ctx->prm.max_range_index = 0;
    if (ctx->prm.additional_mismatch < 1) ctx->prm.additional_mismatch = 1;
    if (ctx->prm.min_close < 1 || ctx->prm.min_close > 64 || ctx->prm.min_perfect_match_around_bp < 0 ||
        ctx->prm.min_perfect_match_around_bp > 64) {
        delete ctx;
        return PG_E_UNSUPPORTED;
    }
    make_tables(ctx);
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0 || p->device < 0 || p->device >= n_dev) {
        // No HIP device: there is deliberately no CPU path behind this ABI.
        delete ctx;
        return PG_E_DEVICE;
    }
    if (hipSetDevice(p->device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return PG_E_DEVICE;
    }
    // The persistent launch spreads its read counters over PG_N_XCD = 8 parts and relies on workgroup b running on XCD
    // b % 8 for locality (never for correctness: a workgroup that finds its part empty moves on to the next).  The
    // MI355X reports 256 compute units in 8 XCDs; on anything else the search still works, the note says what was found.
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) {
            const std::string arch = prop.gcnArchName;
            if (arch.rfind("gfx950", 0) != 0) {
                pg_destroy(ctx);
                return PG_E_DEVICE;                     // the library holds gfx950 code only
            }
            if (prop.multiProcessorCount != 256 && getenv("PG_QUIET") == nullptr)
                fprintf(stderr, "pindel_pg: note: device %d reports %d compute units (an MI355X has 256 in 8 XCDs); the per-XCD "
                                "work split stays correct but was tuned for that shape\n", p->device, prop.multiProcessorCount);
        }
    }
    // the kernel evaluates g_maxMismatch[L] from breakpoints: the table must be monotone (it is for every
    // -e/-E tried).  Whether a batch fits the 16 mismatch levels depends on its longest read and is checked
    // per batch (validate_and_measure).
    for (int i = 1; i < 500; i++)
        if (ctx->mm[i] < ctx->mm[i - 1]) {
            pg_destroy(ctx);
            return PG_E_UNSUPPORTED;
        }
    std::vector<PgLenRec> len_tab(512);
    for (int len = 0; len < 512; len++) len_tab[len] = pg_len_rec(len, ctx->mm, ctx->thr, ctx->prm.additional_mismatch, ctx->prm.min_close);
    if (dev_upload(ctx, &ctx->d_thr, ctx->thr, 512) || dev_upload(ctx, &ctx->d_mm, ctx->mm, 512) ||
        dev_upload(ctx, &ctx->d_len_tab, len_tab.data(), 512)) {
        pg_destroy(ctx);
        return PG_E_DEVICE;
    }
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        g_live_ctx++;
        ctx->counted = true;
    }
    *out = ctx;
    return PG_OK;
}

void pg_destroy(pg_ctx *ctx)
{
    use_device(ctx);
    if (!ctx) return;
    if (ctx->counted) {
        bool last;
        {
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            last = --g_live_ctx == 0;
        }
        if (last) g_pinned.trim();                   // the page-locked result buffers cached for reuse
    }
    free_reference(ctx);
    if (ctx->arena.base) (void)hipFree(ctx->arena.base);
    if (ctx->d_thr) (void)hipFree(ctx->d_thr);
    if (ctx->d_mm) (void)hipFree(ctx->d_mm);
    if (ctx->d_len_tab) (void)hipFree(ctx->d_len_tab);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->dl_stream) (void)hipStreamDestroy(ctx->dl_stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
    delete ctx;
}

const char *pg_last_error(const pg_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int pg_get_max_mismatch(const pg_ctx *ctx, uint32_t *table500)
{
    if (!ctx || !table500) return PG_E_INVALID;
    memcpy(table500, ctx->mm, 500 * sizeof(uint32_t));
    return PG_OK;
}

int pg_load_reference(pg_ctx *ctx, int32_t n_chr, const char *const *names,
                      const uint8_t *const *seq_padded, const uint64_t *len_padded)
{
    use_device(ctx);
    if (!ctx || n_chr <= 0 || !seq_padded || !len_padded) return fail(ctx, PG_E_INVALID, "bad reference arguments");
    if (n_chr > 32767) return fail(ctx, PG_E_UNSUPPORTED, "more than 32767 chromosomes");
    free_reference(ctx);
    uint64_t total_words = 0;
    for (int c = 0; c < n_chr; c++) {
        if (len_padded[c] > PG_MAX_CHR_PADDED) return fail(ctx, PG_E_UNSUPPORTED, "chromosome of 2^31 bases or more (positions are signed 32-bit on the device)");
        if (len_padded[c] < 2ull * ctx->prm.spacer) return fail(ctx, PG_E_INVALID, "chromosome shorter than its spacers");
        total_words += PG_GUARD_WORDS;
        ctx->word_off.push_back(total_words);
        total_words += (len_padded[c] + 31) / 32 + PG_GUARD_WORDS;
        ctx->names.push_back(names && names[c] ? names[c] : "");
        ctx->comp_size.push_back(len_padded[c]);
    }
    total_words += 8;
    try {
        ctx->h_lo.assign(total_words, 0u);
        ctx->h_hi.assign(total_words, 0u);
        ctx->h_nn.assign(total_words, 0xffffffffu);   // guards and tails are N
    } catch (...) {
        free_reference(ctx);
        return fail(ctx, PG_E_NOMEM, "host memory for the packed reference");
    }
    uint8_t code[256];
    memset(code, 4, sizeof code);
    code['A'] = 0; code['C'] = 1; code['G'] = 2; code['T'] = 3;
    for (int c = 0; c < n_chr; c++) {
        const uint8_t *s = seq_padded[c];
        const uint64_t len = len_padded[c];
        uint32_t *lo = &ctx->h_lo[ctx->word_off[c]], *hi = &ctx->h_hi[ctx->word_off[c]],
                 *nn = &ctx->h_nn[ctx->word_off[c]];
        const uint64_t n_words = (len + 31) / 32;
        // words are independent: pack on all host cores (a 3.1 Gbp genome is ~10 s single-threaded)
        auto pack = [&](uint64_t w0, uint64_t w1) {
            for (uint64_t w = w0; w < w1; w++) {
                uint32_t l = 0, h = 0, n = 0xffffffffu;
                const uint64_t b0 = w * 32, cnt = std::min<uint64_t>(32, len - b0);
                for (uint64_t k = 0; k < cnt; k++) {
                    uint8_t cd = code[s[b0 + k]];
                    if (cd < 4) {
                        l |= (uint32_t)(cd & 1u) << k;
                        h |= (uint32_t)(cd >> 1) << k;
                        n &= ~(1u << k);
                    }
                }
                lo[w] = l; hi[w] = h; nn[w] = n;
            }
        };
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned nt = (unsigned)std::min<uint64_t>(hw, std::max<uint64_t>(1, n_words >> 16));
        if (nt <= 1) pack(0, n_words);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back(pack, n_words * t / nt, n_words * (t + 1) / nt);
            for (std::thread &x : th) x.join();
        }
    }
    return upload_reference(ctx);
}

// ---- packed reference on disk: the bit planes exactly as they sit in HBM, so a
// genome is packed from FASTA once (Genome::loadChromosome semantics, pg_load_fasta) and mapped back
// in seconds.  Layout: magic, spacer, n_chr, per chromosome {name length, name, padded size, word
// offset}, number of words, then the lo / hi / N planes.
static const char PG_REF_MAGIC[8] = { 'P', 'G', 'R', 'E', 'F', '0', '1', 0 };

int pg_reference_save_packed(const pg_ctx *ctx, const char *path)
{
    if (!ctx || !path) return PG_E_INVALID;
    if (ctx->names.empty()) return PG_E_NO_REFERENCE;
    FILE *f = fopen(path, "wb");
    if (!f) return PG_E_INVALID;
    bool ok = fwrite(PG_REF_MAGIC, 1, 8, f) == 8;
    const uint32_t spacer = ctx->prm.spacer, n_chr = (uint32_t)ctx->names.size();
    ok = ok && fwrite(&spacer, 4, 1, f) == 1 && fwrite(&n_chr, 4, 1, f) == 1;
    for (uint32_t c = 0; c < n_chr && ok; c++) {
        const uint32_t nl = (uint32_t)ctx->names[c].size();
        ok = fwrite(&nl, 4, 1, f) == 1 && (nl == 0 || fwrite(ctx->names[c].data(), 1, nl, f) == nl) &&
             fwrite(&ctx->comp_size[c], 8, 1, f) == 1 && fwrite(&ctx->word_off[c], 8, 1, f) == 1;
    }
    const uint64_t nw = ctx->h_lo.size();
    ok = ok && fwrite(&nw, 8, 1, f) == 1 && fwrite(ctx->h_lo.data(), 4, nw, f) == nw &&
         fwrite(ctx->h_hi.data(), 4, nw, f) == nw && fwrite(ctx->h_nn.data(), 4, nw, f) == nw;
    ok = (fclose(f) == 0) && ok;
    return ok ? PG_OK : PG_E_INVALID;
}

int pg_reference_load_packed(pg_ctx *ctx, const char *path)
{
    use_device(ctx);
    if (!ctx || !path) return fail(ctx, PG_E_INVALID, "bad packed-reference arguments");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(ctx, PG_E_INVALID, std::string("cannot open ") + path);
    auto bad = [&](const char *why) {
        fclose(f);
        free_reference(ctx);
        return fail(ctx, PG_E_INVALID, std::string(path) + ": " + why);
    };
    free_reference(ctx);
    char magic[8];
    uint32_t spacer = 0, n_chr = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, PG_REF_MAGIC, 8) != 0) return bad("not a packed reference");
    if (fread(&spacer, 4, 1, f) != 1 || fread(&n_chr, 4, 1, f) != 1) return bad("truncated header");
    if (spacer != ctx->prm.spacer) return bad("packed with a different spacer");
    if (n_chr == 0 || n_chr > 32767) return bad("bad chromosome count");
    for (uint32_t c = 0; c < n_chr; c++) {
        uint32_t nl = 0;
        uint64_t size = 0, woff = 0;
        if (fread(&nl, 4, 1, f) != 1 || nl > 4096) return bad("bad chromosome name");
        std::string name(nl, ' ');
        if ((nl && fread(&name[0], 1, nl, f) != nl) || fread(&size, 8, 1, f) != 1 || fread(&woff, 8, 1, f) != 1)
            return bad("truncated chromosome table");
        if (size > PG_MAX_CHR_PADDED || size < 2ull * spacer) return bad("bad chromosome size (2^31 bases or more are not supported)");
        ctx->names.push_back(name);
        ctx->comp_size.push_back(size);
        ctx->word_off.push_back(woff);
    }
    uint64_t nw = 0, expect = 0;
    for (uint32_t c = 0; c < n_chr; c++) {
        expect += PG_GUARD_WORDS;
        if (ctx->word_off[c] != expect) return bad("inconsistent word offsets");
        expect += (ctx->comp_size[c] + 31) / 32 + PG_GUARD_WORDS;
    }
    expect += 8;
    if (fread(&nw, 8, 1, f) != 1 || nw != expect) return bad("inconsistent plane size");
    try {
        ctx->h_lo.resize(nw);
        ctx->h_hi.resize(nw);
        ctx->h_nn.resize(nw);
    } catch (...) {
        fclose(f);
        free_reference(ctx);
        return fail(ctx, PG_E_NOMEM, "host memory for the packed reference");
    }
    if (fread(ctx->h_lo.data(), 4, nw, f) != nw || fread(ctx->h_hi.data(), 4, nw, f) != nw ||
        fread(ctx->h_nn.data(), 4, nw, f) != nw)
        return bad("truncated planes");
    fclose(f);
    return upload_reference(ctx);
}

int pg_load_fasta(pg_ctx *ctx, const char *path)
{
    if (!ctx || !path) return fail(ctx, PG_E_INVALID, "bad fasta arguments");
    std::ifstream in(path, std::ios::binary);
    if (!in) return fail(ctx, PG_E_INVALID, std::string("cannot open ") + path);
    std::string data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    // Genome::loadChromosome (pindel.cpp:272-312): name = first token after '>', every
    // non-blank character is upper-cased, non-ACGT -> N, spacer N's both sides.  The
    // reference's extraction loop appends the final base of the LAST record twice.
    std::vector<std::string> names, seqs;
    size_t i = 0;
    const size_t n = data.size();
    while (i < n && isspace((unsigned char)data[i])) i++;
    if (i >= n || data[i] != '>') return fail(ctx, PG_E_INVALID, "fasta does not start with '>'");
    const std::string spacer(ctx->prm.spacer, 'N');
    while (i < n) {
        i++;   // '>'
        while (i < n && (data[i] == ' ' || data[i] == '\t')) i++;
        size_t j = i;
        while (j < n && !isspace((unsigned char)data[j])) j++;
        names.push_back(data.substr(i, j - i));
        while (j < n && data[j] != '\n') j++;
        std::string s = spacer;
        i = j;
        while (i < n && data[i] != '>') {
            unsigned char ch = (unsigned char)data[i++];
            if (isspace(ch)) continue;
            ch = (unsigned char)toupper(ch);
            if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') ch = 'N';
            s.push_back((char)ch);
        }
        if (i >= n && s.size() > spacer.size()) s.push_back(s.back());
        s += spacer;
        seqs.push_back(std::move(s));
    }
    std::vector<const char *> np;
    std::vector<const uint8_t *> sp;
    std::vector<uint64_t> lp;
    for (size_t c = 0; c < seqs.size(); c++) {
        np.push_back(names[c].c_str());
        sp.push_back((const uint8_t *)seqs[c].data());
        lp.push_back(seqs[c].size());
    }
    return pg_load_reference(ctx, (int32_t)seqs.size(), np.data(), sp.data(), lp.data());
}

int pg_reference_n_chr(const pg_ctx *ctx) { return ctx ? (int)ctx->names.size() : 0; }

const char *pg_reference_name(const pg_ctx *ctx, int32_t c)
{
    return (ctx && c >= 0 && c < (int)ctx->names.size()) ? ctx->names[c].c_str() : nullptr;
}

uint64_t pg_reference_comp_size(const pg_ctx *ctx, int32_t c)
{
    return (ctx && c >= 0 && c < (int)ctx->names.size()) ? ctx->comp_size[c] : 0;
}

int pg_reference_fetch(const pg_ctx *ctx, int32_t c, uint64_t start, uint64_t n, uint8_t *out)
{
    if (!ctx || !out || c < 0 || c >= (int)ctx->names.size()) return PG_E_INVALID;
    if (start + n > ctx->comp_size[c]) return PG_E_INVALID;
    const uint32_t *lo = &ctx->h_lo[ctx->word_off[c]], *hi = &ctx->h_hi[ctx->word_off[c]],
                   *nn = &ctx->h_nn[ctx->word_off[c]];
    for (uint64_t k = 0; k < n; k++) {
        uint64_t p = start + k;
        uint32_t w = (uint32_t)(p >> 5), b = (uint32_t)(p & 31);
        if ((nn[w] >> b) & 1u) out[k] = 'N';
        else out[k] = "ACGT"[((lo[w] >> b) & 1u) | (((hi[w] >> b) & 1u) << 1)];
    }
    return PG_OK;
}

// ---------------------------------------------------------------- device-resident
int pg_device_batch_upload(pg_ctx *ctx, const pg_read_batch *reads, pg_device_batch **out)
{
    use_device(ctx);
    if (!ctx || !out) return PG_E_INVALID;
    *out = nullptr;
    int rc = upload_batch(ctx, reads, out);
    if (rc) return rc;
    return read_exact_count(ctx, *out);
}

static int attach_windows(pg_ctx *ctx, pg_device_batch *b, const pg_windows *bd_hints, bool repack = true);

int pg_device_batch_set_windows(pg_ctx *ctx, pg_device_batch *b, const pg_windows *bd_hints)
{
    use_device(ctx);
    if (!ctx || !b) return PG_E_INVALID;
    int rc = attach_windows(ctx, b, bd_hints);
    if (rc) return rc;
    return read_exact_count(ctx, b);
}

int pg_device_batch_search(pg_ctx *ctx, pg_device_batch *b)
{
    use_device(ctx);
    if (!ctx || !b) return PG_E_INVALID;
    b->modes_done = 0;
    return run_search(ctx, b, PG_MODE_BOTH);
}

int pg_device_batch_pack_search(pg_ctx *ctx, pg_device_batch *b)
{
    use_device(ctx);
    if (!ctx || !b) return PG_E_INVALID;
    b->modes_done = 0;
    int rc = run_search(ctx, b, PG_MODE_BOTH, true);
    if (rc) return rc;
    return b->exact_n < 0 ? read_exact_count(ctx, b) : PG_OK;      // (later searches of the batch skip the exact kernel when its list is empty)
}

int pg_device_batch_repack(pg_ctx *ctx, pg_device_batch *b, double *pack_ms)
{
    use_device(ctx);
    if (!ctx || !b) return PG_E_INVALID;
    if (int stale = stale_batch(ctx, b)) return stale;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    int rc = b->n ? pack_reads(ctx, b, 0, b->n) : PG_OK;
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (pack_ms) *pack_ms = ms;
    return read_exact_count(ctx, b);
}

int pg_device_batch_download(pg_ctx *ctx, pg_device_batch *b, pg_result **out)
{
    use_device(ctx);
    if (!ctx || !b || !out) return PG_E_INVALID;
    pg_result *r = new pg_result();
    int rc = download(ctx, b, r);
    if (rc) {
        delete r;
        return rc;
    }
    *out = r;
    return PG_OK;
}

void pg_device_batch_free(pg_ctx *ctx, pg_device_batch *b)
{
    use_device(ctx);
    (void)ctx;
    if (!b) return;
    free_batch_buffers(b);
    delete b;
}

// Diagnostics (not in the public header): did the last launch that was asked to pack build its records inside the search kernel ?
int pg_debug_last_pack_in_place(const pg_ctx *ctx) { return ctx && ctx->last_in_place ? 1 : 0; }

// Diagnostics (not in the public header): overwrites the batch's packed records and bit planes (tests: what a search that packs in
// place finds afterwards is what it packed itself).
int pg_debug_scribble_records(pg_ctx *ctx, pg_device_batch *b)
{
    use_device(ctx);
    if (!ctx || !b) return PG_E_INVALID;
    const size_t n1 = std::max<size_t>(b->n, 1);
    HIP_TRY(ctx, hipMemsetAsync(b->in_rec, 0xa5, n1 * sizeof(PgInRec), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(b->planes, 0x5a, n1 * 64 * pg_plane_blocks(b->max_len), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

// Diagnostics (not in the public header): raw per-read words of the alg-bytes array.
int pg_debug_read_alg(pg_ctx *ctx, pg_device_batch *b, uint32_t *out, uint32_t n)
{
    use_device(ctx);
    if (!ctx || !b || !out || n > b->n) return PG_E_INVALID;
    {
        int urc = unpack_results(ctx, b);
        if (urc) return urc;
    }
    HIP_TRY(ctx, hipMemcpy(out, b->alg, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PG_OK;
}

// Diagnostics (not in the public header): the `reserved` word of every output record (candidates per read; in a
// -DPG_DIAG build the packed per-read counters).
int pg_debug_read_reserved(pg_ctx *ctx, pg_device_batch *b, uint32_t *out, uint32_t n)
{
    use_device(ctx);
    if (!ctx || !b || !out || n > b->n) return PG_E_INVALID;
    std::vector<PgOutRec> recs(n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n) HIP_TRY(ctx, hipMemcpy(recs.data(), b->out_rec, (size_t)n * sizeof(PgOutRec), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) out[i] = recs[i].reserved;
    return PG_OK;
}

// Diagnostics (-DPG_TIMING builds of the kernel): wave-cycles per phase of the last launch on this batch.
int pg_debug_read_phase_cycles(pg_ctx *ctx, pg_device_batch *b, uint64_t *out, uint32_t n)
{
    use_device(ctx);
    if (!ctx || !b || !out || n > PG_DIAG_WORDS) return PG_E_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, b->pool_used + PG_POOL_SHARDS * 16 + PG_WORK_CTRS * 16, (size_t)n * 8, hipMemcpyDeviceToHost));
    return PG_OK;
}

int pg_last_search_stats(const pg_ctx *ctx, double *kernel_ms, uint64_t *n_runs)
{
    if (!ctx) return PG_E_INVALID;
    if (kernel_ms) *kernel_ms = ctx->last_ms;
    if (n_runs) *n_runs = ctx->last_runs;
    return PG_OK;
}

// Diagnostics: candidates (survivors of the seed filter) the last search of this batch folded, in total.
int pg_device_batch_candidates(pg_ctx *ctx, pg_device_batch *b, double *n_candidates)
{
    use_device(ctx);
    if (!ctx || !b || !n_candidates) return PG_E_INVALID;
    std::vector<PgOutRec> recs(b->n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (b->n) HIP_TRY(ctx, hipMemcpy(recs.data(), b->out_rec, (size_t)b->n * sizeof(PgOutRec), hipMemcpyDeviceToHost));
    double s = 0.0;
    for (const PgOutRec &r : recs) s += r.reserved;
    *n_candidates = s;
    return PG_OK;
}

int pg_device_batch_algorithmic_bytes(pg_ctx *ctx, pg_device_batch *b, double *bytes)
{
    use_device(ctx);
    if (!ctx || !b || !bytes) return PG_E_INVALID;
    {
        int urc = unpack_results(ctx, b);
        if (urc) return urc;
    }
    std::vector<uint32_t> alg(b->n);
    if (b->n) HIP_TRY(ctx, hipMemcpy(alg.data(), b->alg, (size_t)b->n * 4, hipMemcpyDeviceToHost));
    double s = 0.0;
    for (uint32_t v : alg) s += v;
    *bytes = s;
    return PG_OK;
}

// ---------------------------------------------------------------- host in / host out
// Host buffers in, host results out, as a three-stage pipeline over chunks of 64 k to 1 M reads (see `bounds` below):
//   copy stream      the chunk's inputs, host -> HBM
//   compute stream   pack, search (the kernel takes a read range of the batch), delivery kernels: the chunk's runs
//                    gathered in read order behind the earlier chunks', its 64-bit CSR offsets (pg_deliver_chunk)
//   download stream  the chunk's slice of every result array -> pinned host memory, while the next chunk is searched
// The host only waits for "chunk k delivered", reads the chunk's base and run counts (32 bytes) and queues its copies.
static int search_host(pg_ctx *ctx, const pg_read_batch *reads, int mode, pg_result **out)
{
    use_device(ctx);
    if (!ctx || !out) return PG_E_INVALID;
    *out = nullptr;
    pg_device_batch *b = nullptr;
    // (pinned, from the cache: as a malloc'd array the runtime locks its pages for the copy and the free() of 80 MB after a
    // 10 M-read call spent 13 ms in munmap + unpinning)
    HostBuf<uint64_t> off;
    const double t_start = now_ms();
    int rc = alloc_batch(ctx, reads, false, off, &b, true);
    if (rc) return rc;
    double t_alloc = now_ms();
    pg_result *r = nullptr;
    auto bail = [&](int code) {
        // on an error nothing queued on the four streams may outlive the buffers it points at
        if (code) (void)hipDeviceSynchronize();
        free_batch_buffers(b);
        delete b;
        if (code && r) delete r;
        return code;
    };
#define TRY3(call)                                                                           \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return bail(fail(ctx, e_ == hipErrorOutOfMemory ? PG_E_NOMEM : PG_E_DEVICE,      \
                             std::string(#call) + ": " + hipGetErrorString(e_)));            \
    } while (0)
    const uint32_t n = b->n;
    r = new pg_result();
    r->n = n;
    const bool chunk_env = env().host_chunk != 0u;               // (tests: several chunks on a small batch)
    const uint32_t chunk = chunk_env ? env().host_chunk : PG_HOST_CHUNK;
    // Chunk boundaries.  A launch of 256 k reads runs at 278 M reads/s, one of 1 M at ~310, one of 10 M at 323 (ramp-up and
    // tail of the launch itself: profiles/r04/kernel_experiments.txt), but the first chunk's copy and the last chunk's
    // delivery + download are exposed: small chunks first (a quarter of the base chunk, doubling), up to 2^20 reads in the
    // middle, a third of what is left towards the end.
    std::vector<uint32_t> bounds(1, 0u);
    if (n > chunk && !chunk_env) {
        uint64_t ramp = chunk / 4;
        while (bounds.back() < n) {
            const uint64_t left = n - bounds.back();
            const uint64_t mid = std::min<uint64_t>(std::max<uint64_t>(left / 3, chunk), PG_DELIVER_CHUNK);
            bounds.push_back((uint32_t)(bounds.back() + std::min<uint64_t>(std::min(ramp, mid), left)));
            ramp *= 2;
        }
    }
    while (bounds.back() < n) bounds.push_back((uint32_t)std::min<uint64_t>((uint64_t)bounds.back() + chunk, n));
    const uint32_t n_chunks = (uint32_t)bounds.size() - 1;
    // A batch that is ONE chunk (Pindel's own 50 000-read flushes) gets its whole result in ONE device-to-host copy: offsets,
    // summaries and both run lists are laid out in one device block and one pinned host block (pg_result::block), the
    // result's arrays are views into it -- five copies and their ~10 us of runtime call each otherwise.
    const bool single = n_chunks == 1 && !env().no_single_block;
    const size_t cap = single ? 2 * deliver_cap(n) : deliver_cap(n);       // (single: both lists share one buffer)
    size_t o_coff = 0, o_foff = 0, o_rc = 0, o_last = 0, o_max = 0, o_runs = 0, blk_bytes = 0;
    if (single) {
        auto put = [&](size_t bytes) { const size_t at = blk_bytes; blk_bytes += (bytes + 15) & ~(size_t)15; return at; };
        o_coff = put(((size_t)n + 1) * 8);
        o_foff = put(((size_t)n + 1) * 8);
        o_rc = put(n);
        o_last = put((size_t)n * 4);
        o_max = put((size_t)n * 2);
        o_runs = put(cap * sizeof(pg_run));
        if (!r->block.resize(blk_bytes)) return bail(fail(ctx, PG_E_NOMEM, "pinned host memory for the result"));
        uint8_t *h = r->block.data();
        r->close_off.set_view(h + o_coff, (size_t)n + 1, (size_t)n + 1);
        r->far_off.set_view(h + o_foff, (size_t)n + 1, (size_t)n + 1);
        r->rc_flag.set_view(h + o_rc, n, n);
        r->close_last.set_view(h + o_last, n, n);
        r->close_max.set_view(h + o_max, n, n);
        r->close_runs.set_view(h + o_runs, 0, cap);
        r->far_runs.set_view(h + o_runs, 0, 0);                  // (placed behind the close runs once their number is known)
    } else if (!r->close_off.resize((size_t)n + 1) || !r->far_off.resize((size_t)n + 1) || !r->rc_flag.resize(n) ||
               !r->close_last.resize(n) || !r->close_max.resize(n) || !r->close_runs.resize(n ? cap : 0) || !r->far_runs.resize(n ? cap : 0))
        return bail(fail(ctx, PG_E_NOMEM, "pinned host memory for the result"));
    const double t_res = now_ms();
    double t_search = t_res;
    bool whole_batch = false;                // fall back to run_search + download (pool or delivery overflow)
    if (n) {
        if (!ctx->copy_stream) TRY3(hipStreamCreate(&ctx->copy_stream));
        if (!ctx->dl_stream) TRY3(hipStreamCreate(&ctx->dl_stream));
        if (!ctx->stream2) TRY3(hipStreamCreate(&ctx->stream2));
        while (ctx->events.size() < 2 * (size_t)n_chunks + 1) {
            hipEvent_t ev = nullptr;
            TRY3(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ctx->events.push_back(ev);
        }
        // delivery buffers (arena): gathered runs, 64-bit offsets, scratch, per-chunk info; pinned mirror of the info
        pg_run *d_runs[2] = { nullptr, nullptr };
        unsigned long long *d_off[2], *d_tot = b->run_tot, *d_info;
        uint8_t *d_block = nullptr, *d_rc = b->rc_flag;
        uint32_t *d_last = b->close_last;
        uint16_t *d_max = b->close_max;
        void *d_local, *d_blk;
        if (single) {
            if (!(d_block = (uint8_t *)ctx->arena.take(blk_bytes))) return bail(fail(ctx, PG_E_NOMEM, "device arena too small for the delivery buffers"));
            d_off[0] = (unsigned long long *)(d_block + o_coff);
            d_off[1] = (unsigned long long *)(d_block + o_foff);
            d_rc = d_block + o_rc;
            d_last = (uint32_t *)(d_block + o_last);
            d_max = (uint16_t *)(d_block + o_max);
            d_runs[0] = (pg_run *)(d_block + o_runs);            // d_runs[1] stays null: the far runs follow the close runs
        } else if (!(d_runs[0] = (pg_run *)ctx->arena.take(cap * sizeof(pg_run))) || !(d_runs[1] = (pg_run *)ctx->arena.take(cap * sizeof(pg_run))) ||
                   !(d_off[0] = (unsigned long long *)ctx->arena.take(((size_t)n + 1) * 8)) ||
                   !(d_off[1] = (unsigned long long *)ctx->arena.take(((size_t)n + 1) * 8)))
            return bail(fail(ctx, PG_E_NOMEM, "device arena too small for the delivery buffers"));
        if (!(d_local = ctx->arena.take((size_t)PG_DELIVER_CHUNK * 8)) || !(d_blk = ctx->arena.take(4096 * 8)) ||
            !(d_info = (unsigned long long *)ctx->arena.take((size_t)n_chunks * 64)))
            return bail(fail(ctx, PG_E_NOMEM, "device arena too small for the delivery buffers"));
        HostBuf<unsigned long long> info;
        if (!info.resize((size_t)n_chunks * 8)) return bail(fail(ctx, PG_E_NOMEM, "pinned host memory"));
        const uint64_t base0 = reads->seq_off[0];
        const unsigned long long pool_runs = (unsigned long long)b->pool_shard_cap * PG_POOL_SHARDS;
        // (the run-pool cursors, both sets of read counters and the delivery's running totals are zero from alloc_batch's one
        // memset of the output block, queued on ctx->stream)
        TRY3(hipEventRecord(ctx->ev0, ctx->stream));
        if (n_chunks > 1) TRY3(hipEventRecord(ctx->events[2 * n_chunks], ctx->stream));
        hipError_t e = hipSuccess;
        for (uint32_t k = 0; k < n_chunks && e == hipSuccess && rc == PG_OK; k++) {
            const uint32_t lo = bounds[k], hi = bounds[k + 1], cn = hi - lo;
            hipStream_t cs = ctx->copy_stream;
            // A one-chunk batch (Pindel's own 50 000-read flush): the five small arrays lie one after the other in the arena
            // (alloc_batch), so they go as ONE copy from the pinned block that already holds the offsets -- queued before the
            // bases, whose copy from pageable memory blocks the call.  (Five copies cost 60 us of calls and 40 us of serial DMA.)
            const char *d0 = (const char *)b->seq_off;
            const size_t o_str = (size_t)((const char *)b->strand - d0), o_pos = (size_t)((const char *)b->pos - d0),
                         o_isz = (size_t)((const char *)b->isz - d0), o_chr = (size_t)((const char *)b->chr - d0), span = o_chr + (size_t)n * 4;
            const bool one_copy = single && b->in_arena && (const char *)b->strand > d0 && o_str < o_pos && o_pos < o_isz && o_isz < o_chr &&
                                  span <= off.cap_bytes && !env().no_single_block;
            if (one_copy) {
                char *h = (char *)off.data();
                memcpy(h + o_str, reads->anchor_strand, n);
                memcpy(h + o_pos, reads->anchor_pos, (size_t)n * 4);
                memcpy(h + o_isz, reads->insert_size, (size_t)n * 2);
                memcpy(h + o_chr, reads->chr_id, (size_t)n * 4);
                e = hipMemcpyAsync(b->seq_off, h, span, hipMemcpyHostToDevice, cs);
            }
            if (e == hipSuccess && off[hi] > off[lo])
                e = hipMemcpyAsync(b->seq + off[lo], reads->seq + base0 + off[lo], (size_t)(off[hi] - off[lo]), hipMemcpyHostToDevice, cs);
            if (!one_copy) {
                if (e == hipSuccess) e = hipMemcpyAsync(b->seq_off + lo, off.data() + lo, (size_t)(cn + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, cs);
                if (e == hipSuccess) e = hipMemcpyAsync(b->strand + lo, reads->anchor_strand + lo, cn, hipMemcpyHostToDevice, cs);
                if (e == hipSuccess) e = hipMemcpyAsync(b->pos + lo, reads->anchor_pos + lo, (size_t)cn * sizeof(int32_t), hipMemcpyHostToDevice, cs);
                if (e == hipSuccess) e = hipMemcpyAsync(b->isz + lo, reads->insert_size + lo, (size_t)cn * sizeof(int16_t), hipMemcpyHostToDevice, cs);
                if (e == hipSuccess) e = hipMemcpyAsync(b->chr + lo, reads->chr_id + lo, (size_t)cn * sizeof(int32_t), hipMemcpyHostToDevice, cs);
            }
            if (e == hipSuccess) e = hipEventRecord(ctx->events[2 * k], cs);
            // even chunks on one kernel stream, odd chunks on the other: the next chunk's workgroups move in while this
            // chunk's persistent launch drains; the deliveries stay in chunk order (running totals, shared scratch)
            hipStream_t ks = (k & 1u) ? ctx->stream2 : ctx->stream;
            if (e == hipSuccess && k == 1) e = hipStreamWaitEvent(ks, ctx->events[2 * n_chunks], 0);     // (the memsets above)
            if (e == hipSuccess) e = hipStreamWaitEvent(ks, ctx->events[2 * k], 0);
            // (the chunk's records: packed by its search launch itself -- pack in place -- or, 64-bit candidate ids, by a launch of their own)
            if (e == hipSuccess) rc = launch_range(ctx, b, mode, lo, cn, ks, (int)(k & 1u), k < 2, true);
            if (e == hipSuccess && rc == PG_OK) {
                if (k > 0) e = hipStreamWaitEvent(ks, ctx->events[2 * (k - 1) + 1], 0);
                if (e == hipSuccess)
                    e = (hipError_t)pg_deliver_chunk(b->out_rec + lo, cn, d_rc + lo, d_last + lo, d_max + lo, d_local,
                                                     d_blk, d_tot, d_info + 8 * k, b->pool, pool_runs, d_runs[0], d_runs[1], cap, d_off[0] + lo,
                                                     d_off[1] + lo, b->pool_used, ks);
                if (e == hipSuccess) e = hipMemcpyAsync(info.data() + 8 * k, d_info + 8 * k, 64, hipMemcpyDeviceToHost, ks);
                if (e == hipSuccess) e = hipEventRecord(ctx->events[2 * k + 1], ks);
            }
        }
        // (the last delivery waited for every earlier one, so its stream's end is the end of the batch)
        hipStream_t last = ((n_chunks - 1) & 1u) ? ctx->stream2 : ctx->stream;
        if (e == hipSuccess && last != ctx->stream) e = hipStreamWaitEvent(ctx->stream, ctx->events[2 * (n_chunks - 1) + 1], 0);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
        // as the chunks get delivered: their slices to the host
        unsigned long long tot[2] = { 0, 0 };
        for (uint32_t k = 0; k < n_chunks && e == hipSuccess && rc == PG_OK && !whole_batch; k++) {
            const uint32_t lo = bounds[k], hi = bounds[k + 1], cn = hi - lo;
            e = hipEventSynchronize(ctx->events[2 * k + 1]);
            if (e != hipSuccess) break;
            const unsigned long long *in = info.data() + 8 * k;
            if ((single ? in[2] + in[3] > cap : (in[0] + in[2] > cap || in[1] + in[3] > cap)) || in[5] || in[4] > b->pool_shard_cap) {
                whole_batch = true;
                break;
            }
            hipStream_t ds = ctx->dl_stream;
            if (single) {
                // offsets, summaries, close runs, far runs: one copy
                e = hipMemcpyAsync(r->block.data(), d_block, o_runs + (size_t)(in[2] + in[3]) * sizeof(pg_run), hipMemcpyDeviceToHost, ds);
                r->far_runs.set_view(r->block.data() + o_runs + (size_t)in[2] * sizeof(pg_run), (size_t)in[3], (size_t)in[3]);
                tot[0] = in[2];
                tot[1] = in[3];
                break;
            }
            if (in[2]) e = hipMemcpyAsync(r->close_runs.data() + in[0], d_runs[0] + in[0], (size_t)in[2] * sizeof(pg_run), hipMemcpyDeviceToHost, ds);
            if (e == hipSuccess && in[3])
                e = hipMemcpyAsync(r->far_runs.data() + in[1], d_runs[1] + in[1], (size_t)in[3] * sizeof(pg_run), hipMemcpyDeviceToHost, ds);
            if (e == hipSuccess) e = hipMemcpyAsync(r->close_off.data() + lo, d_off[0] + lo, (size_t)cn * 8, hipMemcpyDeviceToHost, ds);
            if (e == hipSuccess) e = hipMemcpyAsync(r->far_off.data() + lo, d_off[1] + lo, (size_t)cn * 8, hipMemcpyDeviceToHost, ds);
            if (e == hipSuccess) e = hipMemcpyAsync(r->rc_flag.data() + lo, b->rc_flag + lo, cn, hipMemcpyDeviceToHost, ds);
            if (mode == PG_MODE_CLOSE) {          // the close-end summary: only a later pg_far_end_batch on this result reads it
                if (e == hipSuccess) e = hipMemcpyAsync(r->close_last.data() + lo, b->close_last + lo, (size_t)cn * 4, hipMemcpyDeviceToHost, ds);
                if (e == hipSuccess) e = hipMemcpyAsync(r->close_max.data() + lo, b->close_max + lo, (size_t)cn * 2, hipMemcpyDeviceToHost, ds);
            }
            tot[0] = in[0] + in[2];
            tot[1] = in[1] + in[3];
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (rc) return bail(rc);
        TRY3(e);
        float ms = 0.f;
        TRY3(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        t_search = now_ms();
        // (the last chunk's info saw every launch's pool cursors: its delivery waited for all the earlier ones)
        const uint32_t worst = (uint32_t)std::min<unsigned long long>(info[8 * (size_t)(n_chunks - 1) + 4], 0xffffffffull);
        const uint64_t total = tot[0] + tot[1];
        if (worst > b->pool_shard_cap) whole_batch = true;
        if (!whole_batch) {
            ctx->last_ms = ms;
            ctx->last_runs = total;
            b->runs_used = total;
            b->modes_done |= mode;
            TRY3(hipStreamSynchronize(ctx->dl_stream));
            r->close_off[n] = tot[0];
            r->far_off[n] = tot[1];
            r->close_runs.resize((size_t)tot[0]);
            r->far_runs.resize((size_t)tot[1]);
            if (mode != PG_MODE_CLOSE) {                  // not downloaded: pg_far_end_batch refuses such a result
                r->close_last.resize(0);
                r->close_max.resize(0);
            }
        } else {
            // a pool shard overflowed, or the lists outgrew the delivery buffers: the whole batch again, with the
            // regrown pool where needed, and the whole-batch download
            TRY3(hipStreamSynchronize(ctx->dl_stream));
            if (worst > b->pool_shard_cap) {
                if ((rc = run_search(ctx, b, mode))) return bail(rc);
            } else {
                b->modes_done |= mode;
                b->runs_used = total;
                ctx->last_ms = ms;
                ctx->last_runs = total;
            }
            b->unpacked = false;
            if ((rc = download(ctx, b, r))) return bail(rc);
        }
    } else {
        r->close_off[0] = r->far_off[0] = 0;
    }
#undef TRY3
    if (g_host_timing)
        fprintf(stderr, "pg_search_batch: %u reads: validate+alloc %.2f ms, result buffers %.2f ms, copy+search+delivery %.2f ms, tail %.2f ms%s\n",
                b->n, t_alloc - t_start, t_res - t_alloc, t_search - t_res, now_ms() - t_search, whole_batch ? " (whole-batch fallback)" : "");
    *out = r;
    return bail(PG_OK);
}

int pg_close_end_batch(pg_ctx *ctx, const pg_read_batch *reads, pg_result **out)
{
    return search_host(ctx, reads, PG_MODE_CLOSE, out);
}

int pg_search_batch(pg_ctx *ctx, const pg_read_batch *reads, pg_result **out)
{
    const double t0 = now_ms();
    const int rc = search_host(ctx, reads, PG_MODE_BOTH, out);
    if (g_host_timing) fprintf(stderr, "pg_search_batch: %.2f ms in all\n", now_ms() - t0);
    return rc;
}

int pg_search_batch_multi(pg_ctx *const *ctxs, int32_t n_ctx, const pg_read_batch *reads, pg_result **out)
{
    if (!ctxs || n_ctx <= 0 || !reads || !out) return PG_E_INVALID;
    for (int k = 0; k < n_ctx; k++)
        if (!ctxs[k]) return PG_E_INVALID;
    *out = nullptr;
    const uint32_t n = reads->n_reads;
    if (n_ctx == 1 || n < 2u * (uint32_t)n_ctx) return search_host(ctxs[0], reads, PG_MODE_BOTH, out);
    // contiguous ranges, one host thread per context; the offsets of a sub-batch need not start at 0
    std::vector<pg_result *> parts((size_t)n_ctx, nullptr);
    std::vector<int> rcs((size_t)n_ctx, PG_OK);
    std::vector<std::thread> th;
    for (int k = 0; k < n_ctx; k++)
        th.emplace_back([&, k]() {
            const uint32_t lo = (uint32_t)((uint64_t)n * k / n_ctx), hi = (uint32_t)((uint64_t)n * (k + 1) / n_ctx);
            pg_read_batch sub = *reads;
            sub.n_reads = hi - lo;
            sub.seq_off += lo;
            sub.anchor_strand += lo;
            sub.anchor_pos += lo;
            sub.insert_size += lo;
            sub.chr_id += lo;
            rcs[(size_t)k] = search_host(ctxs[k], &sub, PG_MODE_BOTH, &parts[(size_t)k]);
        });
    for (std::thread &t : th) t.join();
    int rc = PG_OK;
    for (int k = 0; k < n_ctx; k++)
        if (rcs[(size_t)k]) rc = rcs[(size_t)k];
    pg_result *r = nullptr;
    if (rc == PG_OK) {
        r = new pg_result();
        r->n = n;
        uint64_t tc = 0, tf = 0;
        for (pg_result *p : parts) {
            tc += p->close_runs.size();
            tf += p->far_runs.size();
        }
        if (!r->close_off.resize((size_t)n + 1) || !r->far_off.resize((size_t)n + 1) || !r->rc_flag.resize(n) ||
            !r->close_last.resize(n) || !r->close_max.resize(n) || !r->close_runs.resize(tc) || !r->far_runs.resize(tf)) {
            delete r;
            r = nullptr;
            rc = fail(ctxs[0], PG_E_NOMEM, "pinned host memory for the merged result");
        }
    }
    if (r) {
        uint64_t bc = 0, bf = 0;
        size_t at = 0;
        bool summaries = true;
        for (pg_result *p : parts) {
            const size_t m = p->n;
            for (size_t i = 0; i < m; i++) {
                r->close_off[at + i] = bc + p->close_off[i];
                r->far_off[at + i] = bf + p->far_off[i];
            }
            if (m) {
                memcpy(r->rc_flag.data() + at, p->rc_flag.data(), m);
                if (p->close_last.size() == m && p->close_max.size() == m) {
                    memcpy(r->close_last.data() + at, p->close_last.data(), m * 4);
                    memcpy(r->close_max.data() + at, p->close_max.data(), m * 2);
                } else
                    summaries = false;
            }
            if (p->close_runs.size()) memcpy(r->close_runs.data() + bc, p->close_runs.data(), p->close_runs.size() * sizeof(pg_run));
            if (p->far_runs.size()) memcpy(r->far_runs.data() + bf, p->far_runs.data(), p->far_runs.size() * sizeof(pg_run));
            bc += p->close_runs.size();
            bf += p->far_runs.size();
            at += m;
        }
        r->close_off[n] = bc;
        r->far_off[n] = bf;
        if (!summaries) {
            r->close_last.resize(0);
            r->close_max.resize(0);
        }
    }
    for (pg_result *p : parts) delete p;
    *out = r;
    return rc;
}

// Validates per-read window clusters and uploads them to the batch (replacing earlier ones).  Nothing the reference
// accepts is refused: BDData::getCorrespondingSearchWindowCluster (src/bddata.cpp:949-979) has no cap on the windows of
// a cluster -- clusters of more than 127 windows switch the launch to 64-bit candidate ids (29 bits of window index) --
// and a window of 2^26 positions or more (the width of the position field) is searched as consecutive pieces: the
// reduction over candidates is additive over disjoint position sets.
// repack = false: the caller's next search launch packs the batch itself (launch_range with pack)
static int attach_windows(pg_ctx *ctx, pg_device_batch *b, const pg_windows *bd_hints, bool repack)
{
    const size_t n = b->n;
    if (int stale = stale_batch(ctx, b)) return stale;
    if (b->bd_off) { (void)hipFree(b->bd_off); b->bd_off = nullptr; }
    if (b->bd) { (void)hipFree(b->bd); b->bd = nullptr; }
    b->max_bd_window = 0;
    b->max_bd_cluster = 0;
    if (!(bd_hints && bd_hints->offset && n)) return n && repack ? pack_reads(ctx, b, 0, b->n) : PG_OK;
    const uint64_t nw = bd_hints->offset[n];
    const long long piece = (1ll << PG_REL_BITS) - 1;
    bool split = false;
    for (size_t i = 0; i < n; i++)
        if (bd_hints->offset[i + 1] < bd_hints->offset[i]) return fail(ctx, PG_E_INVALID, "window offsets not monotone");
    if (nw && !bd_hints->windows) return fail(ctx, PG_E_INVALID, "null window array");
    for (uint64_t k = 0; k < nw; k++) {
        const pg_window &w = bd_hints->windows[k];
        if (w.chr_id < 0 || w.chr_id >= (int)ctx->names.size())
            return fail(ctx, PG_E_INVALID, "BreakDancer window on unknown chromosome");
        const long long st = w.start < 0 ? (long long)w.end - 1 : w.start;
        if ((long long)w.end - st > piece) split = true;
    }
    const uint64_t *off = bd_hints->offset;
    const pg_window *win = bd_hints->windows;
    std::vector<uint64_t> off2;
    std::vector<pg_window> win2;
    if (split) {
        off2.reserve(n + 1);
        off2.push_back(0);
        for (size_t i = 0; i < n; i++) {
            for (uint64_t k = bd_hints->offset[i]; k < bd_hints->offset[i + 1]; k++) {
                const pg_window &w = bd_hints->windows[k];
                const long long st = w.start < 0 ? (long long)w.end - 1 : w.start;       // farend_searcher.cpp:69-71
                if ((long long)w.end - st <= piece) {
                    win2.push_back(w);
                    continue;
                }
                for (long long s0 = st; s0 < (long long)w.end; s0 += piece) {
                    pg_window p = { w.chr_id, (int32_t)s0, (int32_t)std::min<long long>(s0 + piece, w.end) };
                    win2.push_back(p);
                }
            }
            off2.push_back(win2.size());
        }
        off = off2.data();
        win = win2.data();
    }
    const uint64_t nw2 = off[n];
    if (nw2 > 0xffffffffull) return fail(ctx, PG_E_UNSUPPORTED, "more than 2^32 BreakDancer windows in a batch");
    for (size_t i = 0; i < n; i++) b->max_bd_cluster = std::max<uint64_t>(b->max_bd_cluster, off[i + 1] - off[i]);
    if (b->max_bd_cluster >= (1ull << 29)) return fail(ctx, PG_E_UNSUPPORTED, "more than 2^29 windows in a BreakDancer cluster");
    for (uint64_t k = 0; k < nw2; k++) {
        const long long st = win[k].start < 0 ? (long long)win[k].end - 1 : win[k].start;
        b->max_bd_window = std::max<int64_t>(b->max_bd_window, (long long)win[k].end - st);
    }
    int rc;
    if ((rc = dev_upload(ctx, &b->bd_off, off, n + 1)) || (rc = dev_upload(ctx, &b->bd, win, (size_t)nw2))) return rc;
    return repack ? pack_reads(ctx, b, 0, b->n) : PG_OK;
}

// Far end of `reads` given each read's close-end summary (host arrays; rc_flag may be null = the sequences are
// already in the orientation GetCloseEnd left them in).  far_off / far_runs of `dst` are replaced.
static int far_end_impl(pg_ctx *ctx, const pg_read_batch *reads, const uint8_t *rc_flag, const uint32_t *close_last,
                        const uint16_t *close_max, const pg_windows *bd_hints, pg_result *dst)
{
    pg_device_batch *b = nullptr;
    HostBuf<uint64_t> off0;
    int rc = alloc_batch(ctx, reads, true, off0, &b, true);
    if (rc) return rc;
    auto bail = [&](int code) {
        free_batch_buffers(b);
        delete b;
        return code;
    };
    const size_t n = b->n;
    if (n) {
        // (the output block of the batch, rc_flag included, was zeroed by alloc_batch; the records are packed by the search launch)
        if ((rc_flag && hipMemcpyAsync(b->rc_flag, rc_flag, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) ||
            hipMemcpyAsync(b->close_last, close_last, n * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(b->close_max, close_max, n * 2, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return bail(fail(ctx, PG_E_DEVICE, "upload of close-end summary failed"));
        const PgSoaOut a = soa_out(b);
        if (pg_pack_close_summary(&a, b->out_rec, b->n, ctx->stream) != 0)
            return bail(fail(ctx, PG_E_DEVICE, "close-end summary pack kernel failed"));
        if (hipStreamSynchronize(ctx->stream) != hipSuccess)       // the host arrays may be pageable temporaries
            return bail(fail(ctx, PG_E_DEVICE, "upload of close-end summary failed"));
    }
    if ((rc = attach_windows(ctx, b, bd_hints, false))) return bail(rc);
    rc = run_search(ctx, b, PG_MODE_FAR, true);
    if (rc) return bail(rc);
    pg_result tmp;
    rc = download(ctx, b, &tmp);
    if (rc) return bail(rc);
    dst->far_off.swap(tmp.far_off);
    dst->far_runs.swap(tmp.far_runs);
    return bail(PG_OK);
}

int pg_far_end_batch(pg_ctx *ctx, const pg_read_batch *reads, pg_result *close, const pg_windows *bd_hints)
{
    use_device(ctx);
    if (!ctx || !reads || !close) return PG_E_INVALID;
    if (close->n != reads->n_reads) return fail(ctx, PG_E_INVALID, "close result does not belong to these reads");
    const size_t n = reads->n_reads;
    if (n && (close->rc_flag.size() != n || close->close_last.size() != n || close->close_max.size() != n))
        return fail(ctx, PG_E_INVALID, "close result carries no close-end summary");
    return far_end_impl(ctx, reads, close->rc_flag.data(), close->close_last.data(), close->close_max.data(), bd_hints, close);
}

int pg_far_end_batch_from_close(pg_ctx *ctx, const pg_read_batch *reads, const uint32_t *close_last,
                                const int16_t *close_max, const pg_windows *bd_hints, pg_result **out)
{
    use_device(ctx);
    if (!ctx || !reads || !out) return PG_E_INVALID;
    *out = nullptr;
    const size_t n = reads->n_reads;
    if (n && (!close_last || !close_max)) return fail(ctx, PG_E_INVALID, "null close-end summary");
    // MaxLenCloseEnd() of a read without UP_Close is 0: such a read is not searched (farend_searcher.cpp:60-66)
    std::vector<uint16_t> cmax(n);
    for (size_t i = 0; i < n; i++) cmax[i] = close_max[i] > 0 ? (uint16_t)close_max[i] : (uint16_t)0;
    pg_result *r = new pg_result();
    r->n = (uint32_t)n;
    if (!r->close_off.assign(n + 1, 0) || !r->rc_flag.assign(n, 0) || !r->close_last.resize(n) || !r->close_max.resize(n)) {
        delete r;
        return fail(ctx, PG_E_NOMEM, "pinned host memory for the result");
    }
    r->close_runs.resize(0);
    for (size_t i = 0; i < n; i++) {
        r->close_last[i] = close_last[i];
        r->close_max[i] = cmax[i];
    }
    int rc = far_end_impl(ctx, reads, nullptr, close_last, cmax.data(), bd_hints, r);
    if (rc) {
        delete r;
        return rc;
    }
    *out = r;
    return PG_OK;
}

int pg_result_view_get(const pg_result *r, pg_result_view *v)
{
    if (!r || !v) return PG_E_INVALID;
    v->n_reads = r->n;
    v->close_off = r->close_off.data();
    v->close_runs = r->close_runs.data();
    v->far_off = r->far_off.data();
    v->far_runs = r->far_runs.data();
    v->rc_flag = r->rc_flag.data();
    return PG_OK;
}

void pg_result_free(pg_result *r) { delete r; }

uint64_t pg_expand_runs(const pg_run *runs, uint64_t n_runs, pg_point *out)
{
    uint64_t k = 0;
    for (uint64_t i = 0; i < n_runs; i++) {
        const pg_run &r = runs[i];
        const bool back = r.flags & PG_RUN_BACKWARD;
        for (uint32_t L = r.len_first; L <= r.len_last; L++, k++) {
            if (!out) continue;
            pg_point &p = out[k];
            uint32_t d = L - r.len_first;
            p.abs_loc = back ? r.abs_loc_first - d : r.abs_loc_first + d;
            p.length = (int16_t)L;
            p.mismatches = r.mismatches;
            p.chr_id = r.chr_id;
            p.direction = back ? '-' : '+';
            p.strand = (r.flags & PG_RUN_ANTISENSE) ? '-' : '+';
        }
    }
    return k;
}

}  // extern "C"
