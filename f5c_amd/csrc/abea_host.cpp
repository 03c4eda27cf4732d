/* abea_host.cpp — the host-buffer entry of libabea_hip.so: abea_align_batch_host (include/abea.h), the replacement
 * of the reference's align_cuda (src/f5c.cu:647-1061) as align_db sees it, plus the multi-device dispatch.
 *
 * The reference flattens the whole batch on one thread into pageable memory (f5c.cu:744-802), copies, runs three
 * kernels, copies back and un-flattens with a reversal loop (f5c.cu:1003-1030), all serialised.  Here:
 *   - reads are ordered longest first (band count E+K) and cut into chunks; a chunk is the unit of flatten -> H2D ->
 *     align-pre + fused align kernel (+ scaling_single) -> D2H -> un-flatten;
 *   - chunks rotate through N slots, each with its own HIP stream, pinned staging and share of the device arena;
 *     kernels of different slots run concurrently on the GPU (the tail of one chunk overlaps the head of the next, so
 *     chunking does not cost the longest-read latency per chunk), copies of both directions overlap kernels;
 *   - the host loops run on a persistent pool of worker threads (the caller's thread takes part): flatten gathers the
 *     event means (4 of event_t's 24 bytes: align.c:131 reads nothing else) and the sequences; un-flatten writes the
 *     caller's per-read buffers;
 *   - what comes down over PCIe is the traceback walk, 2 bits per step (0.4 B per event instead of 8.3): the pairs are
 *     expanded from it on the host while they are written into db->event_align_pairs[i].  ABEA_HOST_PAIRS=device keeps
 *     the expansion on the GPU and copies compacted pair lists instead;
 *   - optional scaling_single (row N1) as the last phase of the alignment kernel: recalibrated scalings, flags and
 *     base_to_event_map come back — the map as one event-count byte per k-mer, rebuilt here — the pair lists need not.
 * With a multi-device context (abea_init_multi) the batch is first split over the devices, longest-processing-time-
 * first on the band count, and each device runs the pipeline above on its share from its own host thread.
 * No CPU alignment fallback exists in this library.
 */
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <iterator>
#include <sched.h>
#include <immintrin.h>
#include "abea_internal.h"

/* ------------------------------------------------------------------ host worker pool */
/* CPUs this process may actually use: min(affinity mask, cgroup CPU quota) */
static int effective_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n > 0 ? n : 1 << 20, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       /* cgroup v2 */
        char q[64]; long long p = 0;
        if (fscanf(f, "%63s %lld", q, &p) == 2 && strcmp(q, "max") != 0 && p > 0)
            n = std::min<long long>(n, std::max<long long>(1, (atoll(q) + p / 2) / p));
        fclose(f);
    } else {                                                                     /* cgroup v1 */
        long long q = -1, p = 0;
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &q) != 1) q = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &p) != 1) p = 0; fclose(g); }
        if (q > 0 && p > 0) n = std::min<long long>(n, std::max<long long>(1, (q + p / 2) / p));
    }
    return std::max(1, n);
}

/* ------------------------------------------------------------------ thread / affinity plan (pure host logic) */
static std::vector<int> parse_cpulist(const char* s) {                      /* "0-3,8,10-11" (sysfs cpulist format) */
    std::vector<int> out;
    if (!s) return out;
    while (*s) {
        while (*s == ',' || *s == ' ' || *s == '\n') ++s;
        if (!*s) break;
        char* e = nullptr;
        long a = strtol(s, &e, 10);
        if (e == s) break;
        long b = a;
        s = e;
        if (*s == '-') { b = strtol(s + 1, &e, 10); s = e; }
        for (long v = a; v <= b && v - a < 65536; ++v) out.push_back((int)v);
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return out;
}

static std::string format_cpulist(const std::vector<int>& v) {
    std::string out;
    for (size_t i = 0; i < v.size();) {
        size_t j = i;
        while (j + 1 < v.size() && v[j + 1] == v[j] + 1) ++j;
        if (!out.empty()) out += ",";
        out += std::to_string(v[i]);
        if (j > i) out += "-" + std::to_string(v[j]);
        i = j + 1;
    }
    return out;
}

/* Worker threads per DEVICE context and the CPUs they bind to.
 *   threads: (usable CPUs - 2) / n_devices, at least 1, at most 16 — two CPUs stay with the HIP runtime's own threads
 *            (under a cgroup CPU quota a pool as wide as the quota gets the whole process throttled); 16
 *            saturate one socket's DRAM bandwidth in the flatten loop.  Round 2 gave ALL devices 16 threads in total
 *            (2 per GPU on an 8-GPU node); the pool is per device now.
 *   binding: the CPUs of the device's NUMA node that the process may run on, when the machine has more than one node
 *            and that set holds at least `threads` CPUs; otherwise unbound.  A device's flatten then reads the caller's
 *            event tables from wherever they are but WRITES its pinned staging (allocated by hipHostMalloc on the node
 *            nearest the current device) locally, and un-flatten reads it locally. */
struct host_thread_plan {
    int threads; std::vector<int> cpus;
    std::vector<std::vector<int>> spread;      /* ABEA_HOST_NUMA=spread: worker t binds to spread[t % size] (one CPU list per NUMA node) */
    std::string key() const { return std::to_string(threads) + "/" + std::to_string(cpus.size()) + "/" + std::to_string(spread.size()); }
};
/* Worker placement (round 5).  DEFAULT = spread: the workers alternate over the NUMA nodes (worker t on node t mod n, free to move
 * among that node's CPUs).  The flatten loop streams the caller's event tables from wherever they lie — on a two-socket host
 * typically both sockets — and an unbound pool lands where the scheduler happens to put it: with most workers on ONE socket every
 * remote line crosses the inter-socket link in the same direction.  Measured on the MI355X host (2 x EPYC, configs[2], same
 * process, alternating): unbound 231 / 268 / 333 ms of flatten per step (and 260-316 ms on other boxes), spread 214 / 215 / 219 ms
 * (profiles/r05/s_numa_spread_ab.txt).  ABEA_HOST_NUMA=0: unbound; =1: every worker on the DEVICE's node (pays only when the
 * caller's tables are there too: 599 ms when they are not); =spread: explicit.  Hosts with one node: unbound. */
static bool host_numa_spread() { const char* nu = getenv("ABEA_HOST_NUMA"); return !nu || !nu[0] || strcmp(nu, "spread") == 0; }
/* ONE switch for the device-node binding, read by the run-time path and by the host-only report alike: opt-in, ABEA_HOST_NUMA=1 */
static bool host_numa_enabled() { const char* nu = getenv("ABEA_HOST_NUMA"); return nu && nu[0] == '1'; }
static std::vector<host_thread_plan> plan_host_threads(int usable_cpus, const std::vector<int>& allowed, int n_devices,
                                                        const int32_t* dev_node, const std::vector<std::vector<int>>& node_cpus,
                                                        int forced_threads, bool numa) {
    std::vector<host_thread_plan> out((size_t)std::max(0, n_devices));
    for (int d = 0; d < n_devices; ++d) {
        int t = usable_cpus > 4 || n_devices > 1 ? (usable_cpus - 2) / n_devices : usable_cpus;
        t = std::max(1, std::min(16, t));
        if (forced_threads > 0) t = forced_threads;
        out[(size_t)d].threads = t;
        const int node = dev_node ? dev_node[d] : -1;
        if (!numa || node < 0 || node >= (int)node_cpus.size() || node_cpus.size() < 2) continue;
        std::vector<int> local;
        if (allowed.empty()) local = node_cpus[(size_t)node];
        else std::set_intersection(node_cpus[(size_t)node].begin(), node_cpus[(size_t)node].end(), allowed.begin(), allowed.end(),
                                   std::back_inserter(local));
        if ((int)local.size() >= t) out[(size_t)d].cpus = local;
    }
    return out;
}

/* host-only entry over the same rule, so that the plan can be tested without a GPU (cpulists in sysfs format) */
extern "C" int abea_host_plan_threads(int32_t usable_cpus, const char* allowed_cpulist, int32_t n_devices,
                                      const int32_t* device_numa_node, int32_t n_nodes, const char* const* node_cpulist,
                                      int32_t* threads_per_device, char* bind_cpulists, size_t cap_each) {
    if (usable_cpus < 1 || n_devices < 1 || !threads_per_device || n_nodes < 0 || (n_nodes && !node_cpulist) ||
        (bind_cpulists && cap_each < 2))
        return abea_fail(ABEA_EINVAL, "abea_host_plan_threads: bad argument");
    std::vector<std::vector<int>> nodes;
    for (int32_t i = 0; i < n_nodes; ++i) nodes.push_back(parse_cpulist(node_cpulist[i]));
    const char* e = getenv("ABEA_HOST_THREADS");
    const std::vector<host_thread_plan> plan = plan_host_threads(usable_cpus, parse_cpulist(allowed_cpulist), n_devices, device_numa_node,
                                                                 nodes, e ? std::max(1, atoi(e)) : 0, host_numa_enabled());
    for (int32_t d = 0; d < n_devices; ++d) {
        threads_per_device[d] = plan[(size_t)d].threads;
        if (bind_cpulists) snprintf(bind_cpulists + (size_t)d * cap_each, cap_each, "%s", format_cpulist(plan[(size_t)d].cpus).c_str());
    }
    return ABEA_OK;
}

static void bind_this_thread(const std::vector<int>& cpus) {
    if (cpus.empty()) return;
    cpu_set_t set; CPU_ZERO(&set);
    for (int cpu : cpus) if (cpu >= 0 && cpu < CPU_SETSIZE) CPU_SET(cpu, &set);
    sched_setaffinity(0, sizeof set, &set);            /* best effort: a failure leaves the thread where it was */
}

/* the machine as the plan sees it: the process's affinity mask and the CPU list of every NUMA node */
static std::vector<int> allowed_cpus() {
    std::vector<int> out;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0)
        for (int i = 0; i < CPU_SETSIZE; ++i) if (CPU_ISSET(i, &set)) out.push_back(i);
    return out;
}
static std::vector<std::vector<int>> numa_node_cpus() {
    /* indexed by node number; the nodes that exist are listed in .../node/online (numbering may be sparse: a missing node
     * stays an empty list instead of ending the enumeration) */
    std::vector<int> online;
    if (FILE* f = fopen("/sys/devices/system/node/online", "r")) {
        char buf[1024];
        const size_t got = fread(buf, 1, sizeof buf - 1, f);
        buf[got] = 0;
        fclose(f);
        online = parse_cpulist(buf);
    }
    std::vector<std::vector<int>> out;
    for (int n : online) {
        if (n < 0 || n >= 1024) continue;
        char path[96], buf[4096];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", n);
        FILE* f = fopen(path, "r");
        if (!f) continue;
        const size_t got = fread(buf, 1, sizeof buf - 1, f);
        buf[got] = 0;
        fclose(f);
        if ((int)out.size() <= n) out.resize((size_t)n + 1);
        out[(size_t)n] = parse_cpulist(buf);
    }
    return out;
}
/* NUMA node of a HIP device: /sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node; -1 when unknown */
int abea_device_numa_node(int device) {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return -1;
    for (char* p = bdf; *p; ++p) *p = (char)tolower(*p);
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    return node;
}

struct abea_host_pool {
    typedef std::function<void(int64_t, int64_t)> fn_t;       /* fn(lo, hi): items [lo, hi) */
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const fn_t* fn = nullptr;
    std::atomic<int64_t> next{0};
    int64_t n = 0, grain = 1;
    int running = 0;
    uint64_t gen = 0;
    bool stop = false;

    /* `cpus` (may be empty): the CPUs the workers bind to — the NUMA node of the pool's device (plan_host_threads) */
    std::string key;                                          /* of the plan it was built for (threads + binding) */
    explicit abea_host_pool(int threads, const std::vector<int>& cpus = std::vector<int>(),
                            const std::vector<std::vector<int>>& spread = std::vector<std::vector<int>>()) {
        for (int t = 1; t < threads; ++t) {
            const std::vector<int> mine = spread.empty() ? cpus : spread[(size_t)t % spread.size()];
            th.emplace_back([this, mine]() { bind_this_thread(mine); worker(); });
        }
    }
    ~abea_host_pool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_go.notify_all();
        for (auto& t : th) t.join();
    }
    int threads() const { return (int)th.size() + 1; }
    void drain() {
        for (;;) {
            const int64_t lo = next.fetch_add(grain);
            if (lo >= n) break;
            (*fn)(lo, std::min(n, lo + grain));
        }
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&]() { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
            }
            drain();
            { std::lock_guard<std::mutex> lk(mu); if (--running == 0) cv_done.notify_all(); }
        }
    }
    /* run f over [0, n_items) in pieces of `g` items, dynamically scheduled; the caller's thread takes part */
    void run(int64_t n_items, int64_t g, const fn_t& f) {
        if (n_items <= 0) return;
        if (th.empty() || n_items <= g) { f(0, n_items); return; }
        {
            std::lock_guard<std::mutex> lk(mu);
            fn = &f; n = n_items; grain = std::max<int64_t>(1, g); next.store(0);
            running = (int)th.size(); ++gen;
        }
        cv_go.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&]() { return running == 0; });
        fn = nullptr;
    }
};

/* ------------------------------------------------------------------ one chunk in flight */
struct abea_host_slot {
    hipStream_t stream = nullptr;
    hipStream_t stream_hi = nullptr;                    /* highest priority: what follows the alignment kernel of a chunk */
    bool hi_always = false, hi_never = false;
    hipStream_t post = nullptr;                         /* the stream the chunk in flight ends on */
    hipEvent_t k0 = nullptr, k1 = nullptr, k2 = nullptr, kdone = nullptr, done = nullptr;
    uint8_t* up = nullptr;  size_t up_cap = 0;          /* pinned, host -> device: [desc][reads][evm] */
    uint8_t* dn = nullptr;  size_t dn_cap = 0;          /* pinned, device -> host */
    bool busy = false;
    /* ---- the chunk in flight ---- */
    int32_t m = 0, chunk_no = 0;
    std::vector<int32_t> rd;                            /* caller index of descriptor j */
    struct read_offs { plan_offsets o; int64_t read_off, pair_off, kmer_off; int32_t src; };
    std::vector<read_offs> offs;                        /* the serial part of a chunk's plan: where descriptor j's arrays start */
    bool scaling = false, device_pairs = false, staged = false;
    size_t o_np = 0, o_diag = 0, o_codes = 0, o_poff = 0, o_cursor = 0, o_pairs = 0;      /* offsets in `dn` */
    size_t o_sc = 0, o_epb = 0, o_flag = 0, o_nal = 0, o_cnt = 0, o_var = 0;
    abea_pair_t* d_pairs = nullptr;                     /* device-pairs mode: compacted lists to copy at stage A */
    size_t pair_cap = 0;
};

/* the slot's high-priority stream, created when first needed */
static int slot_hi_stream(abea_host_slot& s, hipStream_t* out) {
    if (s.hi_never) { *out = s.stream; return ABEA_OK; }
    if (!s.stream_hi) {
        int prio_lo = 0, prio_hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIP_TRY(hipStreamCreateWithPriority(&s.stream_hi, hipStreamNonBlocking, prio_hi));
    }
    *out = s.stream_hi;
    return ABEA_OK;
}

static int slot_create(abea_host_slot** out) {
    abea_host_slot* s = new abea_host_slot();
    *out = s;
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    /* The kernels behind a chunk's alignment (scaling_single, the copy-out of the result block) are small; on the slot's own
     * stream their workgroups queue for wave slots behind the alignment kernels of the OTHER slots, which hold all 4096 of
     * them for milliseconds at a time (measured: 25 ms per chunk for 1 ms of work).  A queue of higher priority is served first
     * whenever slots free up. */
    const char* hi = getenv("ABEA_HOST_HI_STREAM");           /* "1": the copy-out of a chunk goes through a high-priority stream; default: the slot's own */
    s->hi_never = hi && hi[0] == '0';
    s->hi_always = hi && hi[0] == '1';
    HIP_TRY(hipEventCreate(&s->k0)); HIP_TRY(hipEventCreate(&s->k1)); HIP_TRY(hipEventCreate(&s->k2));
    HIP_TRY(hipEventCreateWithFlags(&s->kdone, hipEventDisableTiming | (getenv("ABEA_HOST_SPIN") ? 0 : hipEventBlockingSync)));
    HIP_TRY(hipEventCreateWithFlags(&s->done, hipEventDisableTiming | (getenv("ABEA_HOST_SPIN") ? 0 : hipEventBlockingSync)));
    return ABEA_OK;
}

int abea_default_host_threads() {
    const char* e = getenv("ABEA_HOST_THREADS");
    return plan_host_threads(effective_cpus(), std::vector<int>(), 1, nullptr, std::vector<std::vector<int>>(),
                             e ? std::max(1, atoi(e)) : 0, false)[0].threads;
}

/* ------------------------------------------------------------------ lanes: what one batch in flight owns */
/* A lane is a disjoint set of stream slots, a share of the device arena and a worker pool.  The synchronous entry runs
 * on the FULL lane (all slots, the whole arena, the whole pool); abea_align_batch_host_submit() runs each batch on one
 * of n_lanes equal lanes, so that several host batches are in flight on one device at once (f5c's default batches are
 * latency-bound: one lasts as long as its longest read and fills an eighth of the wave slots). */
struct abea_host_lane {
    int first_slot = 0, n_slots = 0;
    size_t arena_off = 0, arena_bytes = 0;
    abea_host_pool* pool = nullptr;
};

struct abea_async_job {
    std::thread th;
    abea_host_batch H;
    abea_stats st;
    int rc = ABEA_OK;
    std::string err;
    bool active = false;
    bool waiting = false;                      /* a thread is inside abea_align_batch_host_wait() for this job */
    uint32_t gen = 0;
};

struct abea_host_async {
    std::vector<host_thread_plan> plan; std::string plan_key = "?";     /* context_thread_plan's cache (top-level context) */
    std::mutex plan_mu;                        /* ... which the lanes' threads of submitted batches consult at the same moment */
    int n_lanes = 2;
    abea_host_lane full;                       /* per device context */
    std::vector<abea_host_lane> lanes;         /* per device context */
    abea_async_job jobs[ABEA_MAX_INFLIGHT];    /* top-level context only */
    int n_active = 0;                          /* top-level context only */
};

static abea_host_async* async_of(abea_ctx* c) {
    if (!c->async) c->async = new abea_host_async();
    return c->async;
}

/* the plan for the devices of a context (children of a multi-device parent, or the context itself) */
static std::vector<host_thread_plan> context_thread_plan(abea_ctx* c) {
    std::vector<int32_t> nodes;
    if (c->children.empty()) nodes.push_back(c->numa_node);
    else for (abea_ctx* ch : c->children) nodes.push_back(ch->numa_node);
    const char* e = getenv("ABEA_HOST_THREADS");
    /* the plan is cached per context: it was re-derived (sched_getaffinity + up to 64 sysfs reads) on every batch, which is
     * per-call overhead on the latency-bound default f5c batches (round-3 advisor finding); the key is the two switches */
    const bool numa = host_numa_enabled();
    abea_host_async* a = async_of(c);
    const std::string key = std::string(e ? e : "") + "|" + (numa ? "1" : host_numa_spread() ? "s" : "0");
    /* every batch in flight runs this on its own thread: the first batches after abea_init all find the cache empty and would
     * fill it at once (round 6: one GPU suite in a dozen died here with a segmentation fault — a vector read while another
     * thread re-assigned it) */
    std::lock_guard<std::mutex> lk(a->plan_mu);
    if (a->plan_key == key && a->plan.size() == nodes.size()) return a->plan;
    /* Binding is OPT-IN at run time (ABEA_HOST_NUMA=1).  Measured on the MI355X box (2 sockets, bench.py, 100 k reads): with
     * the workers of the one device bound to its node, flatten went from 217 to 602 ms per step — the loop READS 24 B per
     * event from the caller's tables, which live wherever the caller's threads first touched them (both nodes), and only
     * WRITES 4 B per event to the local pinned staging; binding trades the small local write for remote reads through one
     * socket.  It pays only when the caller places each device's share of the batch on that device's node. */
    a->plan = plan_host_threads(effective_cpus(), numa ? allowed_cpus() : std::vector<int>(), (int)nodes.size(), nodes.data(),
                                numa ? numa_node_cpus() : std::vector<std::vector<int>>(), e ? std::max(1, atoi(e)) : 0, numa);
    if (host_numa_spread()) {                                  /* one CPU list per NUMA node the process may run on */
        const std::vector<int> allowed = allowed_cpus();
        std::vector<std::vector<int>> per_node;
        for (const std::vector<int>& nc : numa_node_cpus()) {
            std::vector<int> local;
            std::set_intersection(nc.begin(), nc.end(), allowed.begin(), allowed.end(), std::back_inserter(local));
            if (!local.empty()) per_node.push_back(local);
        }
        if (per_node.size() >= 2) for (host_thread_plan& pl : a->plan) pl.spread = per_node;
    }
    a->plan_key = key;
    return a->plan;
}

int abea_host_batches_in_flight(abea_ctx* c) { return c->async ? c->async->n_active : 0; }

static void lanes_release(abea_ctx* c) {
    if (!c->async) return;
    for (abea_host_lane& l : c->async->lanes) delete l.pool;
    c->async->lanes.clear();
    delete c->async->full.pool;
    c->async->full.pool = nullptr;
}

void abea_parallel_for(abea_ctx* c, int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)>& f) {
    abea_host_async* a = async_of(c);
    if (!a->full.pool) {
        const host_thread_plan pl = context_thread_plan(c)[0];
        a->full.pool = new abea_host_pool(pl.threads, pl.cpus, pl.spread);
        a->full.pool->key = pl.key();
    }
    a->full.pool->run(n, grain, f);
}

void abea_host_join_async(abea_ctx* c) {
    if (!c->async) return;
    for (abea_async_job& j : c->async->jobs) if (j.th.joinable()) j.th.join();
}

void abea_host_release(abea_ctx* c) {
    for (abea_host_slot* s : c->slots) {
        if (!s) continue;
        if (s->stream_hi && s->stream_hi != s->stream) { hipStreamSynchronize(s->stream_hi); hipStreamDestroy(s->stream_hi); }
        if (s->stream) { hipStreamSynchronize(s->stream); hipStreamDestroy(s->stream); }
        for (hipEvent_t e : {s->k0, s->k1, s->k2, s->kdone, s->done}) if (e) hipEventDestroy(e);
        hipHostFree(s->up); hipHostFree(s->dn);
        delete s;
    }
    c->slots.clear();
    if (c->async) {
        for (abea_async_job& j : c->async->jobs) if (j.th.joinable()) j.th.join();
        lanes_release(c);
        delete c->async;
        c->async = nullptr;
    }
}

/* ------------------------------------------------------------------ un-flatten helpers */
/* The walk's 2-bit codes -> (k-mer, event) pairs in ascending order (align.c:452-513: the reference collects the
 * pairs backwards, then reverses).  Step j of the walk (j = 0 at the end cell (K-1, best_event)) is pair n-1-j;
 * code 0 = diagonal (k-mer and event step), 1 = up (event only), 2 = left (k-mer only) — the same expansion as
 * phase 3 of abea_align_kernel. */
static inline uint32_t walk_code(const uint32_t* codes, int32_t t) { return (codes[t >> 4] >> ((t & 15) * 2)) & 3u; }

/* 16 steps at a time with AVX2 + BMI2 (every x86 host an MI355X sits in has them; checked at run time): the two decrement
 * bits of each step (k-mer steps unless the code is 1, event steps unless it is 2) come out of the word with pext, their
 * exclusive prefix counts with four byte-shift adds, and the 16 pairs leave as four 32-byte non-temporal stores instead of
 * sixteen 8-byte ones: the un-flatten loop is bound by the store rate of the few worker threads a cgroup quota leaves
 * (round 4; 1.2 -> 0.45 ns per pair on one core of the build box). */
__attribute__((target("avx2,bmi2")))
static inline __m128i walk_prefix16(uint32_t bits, __m128i rev) {                         /* byte s = steps before s that moved */
    const __m128i x = _mm_set_epi64x((long long)_pdep_u64(bits >> 8, 0x0101010101010101ull),
                                     (long long)_pdep_u64(bits & 0xFFu, 0x0101010101010101ull));
    __m128i p = _mm_add_epi8(x, _mm_slli_si128(x, 1));
    p = _mm_add_epi8(p, _mm_slli_si128(p, 2));
    p = _mm_add_epi8(p, _mm_slli_si128(p, 4));
    p = _mm_add_epi8(p, _mm_slli_si128(p, 8));
    return _mm_shuffle_epi8(_mm_sub_epi8(p, x), rev);                                      /* exclusive, step 15 first */
}

__attribute__((target("avx2,bmi2")))
static void expand_codes_avx2(const uint32_t* codes, int32_t n, int32_t k, int32_t e, abea_pair_t* out) {
    int32_t t = 0;
    /* scalar steps until the end of the block to be written is 32-byte aligned (the lists are written back to front) */
    for (; t < n && (reinterpret_cast<uintptr_t>(out + (n - t)) & 31u); ++t) {
        _mm_stream_si64(reinterpret_cast<long long*>(out + (n - 1 - t)), (long long)(((uint64_t)(uint32_t)e << 32) | (uint32_t)k));
        const uint32_t cd = walk_code(codes, t);
        k -= (cd != 1u); e -= (cd != 2u);
    }
    const __m128i rev = _mm_set_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    for (; t + 16 <= n; t += 16) {
        const int32_t idx = t >> 4, sh = (t & 15) * 2;
        const uint32_t w = sh ? (codes[idx] >> sh) | (codes[idx + 1] << (32 - sh)) : codes[idx];
        const uint32_t b0 = _pext_u32(w, 0x55555555u), b1 = _pext_u32(w, 0xAAAAAAAAu);
        const uint32_t dkb = ~(b0 & ~b1) & 0xFFFFu, deb = ~(b1 & ~b0) & 0xFFFFu;          /* bit s: step s moves k / e */
        const __m128i pk = walk_prefix16(dkb, rev), pe = walk_prefix16(deb, rev);
        const __m256i kv = _mm256_set1_epi32(k), ev = _mm256_set1_epi32(e);
        __m256i* o = reinterpret_cast<__m256i*>(out + (n - t - 16));                       /* 32-byte aligned by the peel above */
        for (int h = 0; h < 2; ++h) {                                                      /* steps 15..8, then 7..0 */
            const __m256i k8 = _mm256_sub_epi32(kv, _mm256_cvtepu8_epi32(h ? _mm_srli_si128(pk, 8) : pk));
            const __m256i e8 = _mm256_sub_epi32(ev, _mm256_cvtepu8_epi32(h ? _mm_srli_si128(pe, 8) : pe));
            const __m256i lo = _mm256_unpacklo_epi32(k8, e8), hi = _mm256_unpackhi_epi32(k8, e8);   /* {ref_pos, read_pos} */
            _mm256_stream_si256(o + 2 * h, _mm256_permute2x128_si256(lo, hi, 0x20));
            _mm256_stream_si256(o + 2 * h + 1, _mm256_permute2x128_si256(lo, hi, 0x31));
        }
        k -= __builtin_popcount(dkb); e -= __builtin_popcount(deb);
    }
    for (; t < n; ++t) {
        _mm_stream_si64(reinterpret_cast<long long*>(out + (n - 1 - t)), (long long)(((uint64_t)(uint32_t)e << 32) | (uint32_t)k));
        const uint32_t cd = walk_code(codes, t);
        k -= (cd != 1u); e -= (cd != 2u);
    }
    _mm_sfence();
}

static void expand_codes(const uint32_t* codes, int32_t n, int32_t k, int32_t e, abea_pair_t* out) {
    static const bool simd = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !getenv("ABEA_HOST_SCALAR_EXPAND");
    if (simd) { expand_codes_avx2(codes, n, k, e, out); return; }
    /* the caller's pair buffer is written once and not read back here: 8-byte non-temporal stores (the write-combining
     * buffers assemble full lines back to front) halve the DRAM traffic of a plain store's read-for-ownership */
    long long* o = reinterpret_cast<long long*>(out + n);
    for (int32_t j = 0; j < n; j += 16) {
        uint32_t w = codes[j >> 4];
        const int32_t lim = std::min(16, n - j);
        for (int32_t t = 0; t < lim; ++t) {
            --o;
            _mm_stream_si64(o, (long long)(((uint64_t)(uint32_t)e << 32) | (uint32_t)k));      /* {ref_pos = k, read_pos = e} */
            const uint32_t cd = w & 3u;
            w >>= 2;
            k -= (cd != 1u); e -= (cd != 2u);
        }
    }
    _mm_sfence();
}

/* The same walk -> db->base_to_event_map[i] (postalign, align.c:571-596), without going through the pair list: pair i is
 * a NEW event iff its predecessor in the list has a different event, i.e. iff the step leaving it is not a "left"
 * (k-mer only) step; a k-mer's entry is {first, last} new event of its run of pairs, {-1,-1} when the run has none.
 * Every k-mer from the first pair's to the last pair's has a run (a step changes the k-mer by at most one); k-mers below
 * the first pair's stay {-1,-1} (none for a list that passed QC: it spans k-mer 0, align.c:529). */
static void expand_codes_to_map(const uint32_t* codes, int32_t n, int32_t k, int32_t e, abea_index_pair_t* map) {
    /* One iteration per K-MER, not per step (a first version walked the steps with two data-dependent branches each and
     * took 740 ms per 100 k reads against 110 ms for the pair expansion).  In walk order a k-mer's pairs are (k, e_hi),
     * (k, e_hi - 1), ..., (k, e_lo), joined by "up" steps (code 1) and left by a diagonal (0) or a "left" (2) step.  All but
     * the last are new events; the last is new iff it is left diagonally (or is pair 0).  So
     *   left diagonally:  {e_lo, e_hi}      left by a skip:  {e_lo + 1, e_hi} if the run has an up step, else {-1, -1}
     * and the next k-mer starts at e_lo - 1 (diagonal) or e_lo (skip).  Runs of code 1 are counted with tzcnt over a mask
     * of the non-1 codes, 32 codes per 64-bit word. */
    long long* m64 = reinterpret_cast<long long*>(map);
    const uint64_t ONES = 0x5555555555555555ull;
    int32_t nu = 0;                                          /* up steps of the current k-mer so far */
    for (int32_t j0 = 0; j0 < n; j0 += 32) {
        const int32_t cnt = std::min(32, n - j0);
        uint64_t w = (uint64_t)codes[j0 >> 4];
        if (cnt > 16) w |= (uint64_t)codes[(j0 >> 4) + 1] << 32;
        if (j0 + cnt == n) w &= ~(3ull << (2 * (cnt - 1)));  /* the last step's code is never applied: pair 0 is new, like a diagonal leave */
        const uint64_t x = w ^ ONES;
        uint64_t non_up = (x | (x >> 1)) & ONES;             /* bit 2t: code t is not "up" */
        if (cnt < 32) non_up &= (1ull << (2 * cnt)) - 1;
        int32_t pos = 0;
        while (non_up) {
            const int32_t t = (int32_t)(__builtin_ctzll(non_up) >> 1);
            nu += t - pos;
            const uint32_t skip = (uint32_t)(w >> (2 * t + 1)) & 1u;      /* code 2 = left by a skip, 0 = diagonally */
            const int32_t e_lo = e - nu;
            const int32_t start = skip ? (nu ? e_lo + 1 : -1) : e_lo;
            const int32_t stop = (skip && !nu) ? -1 : e;
            _mm_stream_si64(m64 + k, (long long)(((uint64_t)(uint32_t)stop << 32) | (uint32_t)start));
            --k; e = e_lo - (int32_t)(skip ^ 1u);
            nu = 0; pos = t + 1;
            non_up &= non_up - 1;
        }
        nu += cnt - pos;                                     /* trailing up steps run on into the next word */
    }
    for (; k >= 0; --k) _mm_stream_si64(m64 + k, -1ll);     /* k-mers below the first pair's: {-1, -1} */
    _mm_sfence();
}

/* The same map from the per-k-mer event counts the fused scaling_single phase leaves (abea_fused_scaling.kcnt): the entries of
 * base_to_event_map tile the events of the path in k order — every event is NEW for exactly one k-mer (postalign, align.c:571-596)
 * and a k-mer's entry is {first, last} new event of its run — so entry k is the `count[k]` events ending where entry k + 1's
 * begin, the last non-empty one ending at the alignment's end event.  Two adds and a store per k-mer (the walk needs a tzcnt
 * chain with a serial dependency: 142 ms per 100 k reads against 110 ms for the pair lists; this: see DESIGN.md §6).  Returns false
 * at a count of 255 ("255 or more"): the caller rebuilds that read's map from the walk. */
/* eight entries at a time (AVX2): inclusive prefix sums P of the eight counts in 32-bit lanes; with `top` the last event at or
 * below the block's highest k-mer, entry j is {stop - c + 1, stop}, stop = top - total + P[j], or {-1, -1} when c = 0 */
__attribute__((target("avx2")))
static bool expand_counts_to_map_avx2(const uint8_t* count, int32_t K, int32_t end_event, abea_index_pair_t* map) {
    long long* m64 = reinterpret_cast<long long*>(map);
    int32_t cur = end_event, k = K - 1;
    auto one = [&](int32_t kk) {                         /* scalar step: head (until the block stores are 32-byte aligned) and tail */
        const int32_t c = count[kk];
        if (c == 255) return false;
        const int32_t start = c ? cur - c + 1 : -1, stop = c ? cur : -1;
        _mm_stream_si64(m64 + kk, (long long)(((uint64_t)(uint32_t)stop << 32) | (uint32_t)start));
        cur -= c;
        return true;
    };
    for (; k >= 0 && (reinterpret_cast<uintptr_t>(map + k + 1) & 31u); --k) if (!one(k)) return false;
    const __m256i ones = _mm256_set1_epi32(1), neg = _mm256_set1_epi32(-1), zero = _mm256_setzero_si256();
    const __m256i lane3 = _mm256_set1_epi32(3), lane7 = _mm256_set1_epi32(7);
    __m256i curv = _mm256_set1_epi32(cur);               /* the running event stays in a vector: the loop-carried chain is one
                                                          * permute and one subtract per eight entries */
    for (; k >= 7; k -= 8) {
        const __m128i c8 = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(count + k - 7));
        if (_mm_movemask_epi8(_mm_cmpeq_epi8(c8, _mm_set1_epi8((char)255))) & 0xFF) return false;
        const __m256i v = _mm256_cvtepu8_epi32(c8);
        __m256i p = _mm256_add_epi32(v, _mm256_slli_si256(v, 4));          /* prefix sums inside each 128-bit half ... */
        p = _mm256_add_epi32(p, _mm256_slli_si256(p, 8));
        const __m256i low_total = _mm256_permutevar8x32_epi32(p, lane3);
        p = _mm256_add_epi32(p, _mm256_blend_epi32(zero, low_total, 0xF0));   /* ... the upper half continues the lower one */
        curv = _mm256_sub_epi32(curv, _mm256_permutevar8x32_epi32(p, lane7)); /* the event below the block, in every lane */
        const __m256i stop = _mm256_add_epi32(curv, p);
        const __m256i start = _mm256_add_epi32(_mm256_sub_epi32(stop, v), ones);
        const __m256i none = _mm256_cmpeq_epi32(v, zero);
        const __m256i st = _mm256_blendv_epi8(start, neg, none), sp = _mm256_blendv_epi8(stop, neg, none);
        const __m256i lo = _mm256_unpacklo_epi32(st, sp), hi = _mm256_unpackhi_epi32(st, sp);       /* {start, stop} */
        __m256i* o = reinterpret_cast<__m256i*>(map + k - 7);                /* 32-byte aligned by the head loop */
        _mm256_stream_si256(o, _mm256_permute2x128_si256(lo, hi, 0x20));
        _mm256_stream_si256(o + 1, _mm256_permute2x128_si256(lo, hi, 0x31));
    }
    cur = _mm256_extract_epi32(curv, 0);
    for (; k >= 0; --k) if (!one(k)) return false;
    _mm_sfence();
    return true;
}

static bool expand_counts_to_map(const uint8_t* count, int32_t K, int32_t end_event, abea_index_pair_t* map) {
    static const bool simd = __builtin_cpu_supports("avx2") && !getenv("ABEA_HOST_SCALAR_EXPAND");
    if (simd) return expand_counts_to_map_avx2(count, K, end_event, map);
    long long* m64 = reinterpret_cast<long long*>(map);
    int32_t cur = end_event;
    for (int32_t k = K - 1; k >= 0; --k) {
        const int32_t c = count[k];
        if (c == 255) return false;
        const int32_t start = c ? cur - c + 1 : -1, stop = c ? cur : -1;
        _mm_stream_si64(m64 + k, (long long)(((uint64_t)(uint32_t)stop << 32) | (uint32_t)start));
        cur -= c;
    }
    _mm_sfence();
    return true;
}

/* host-only entry over the same routine (unit tests); returns ABEA_EINVAL for a table that holds the escape value */
extern "C" int abea_expand_kmer_counts_to_map(const uint8_t* count, int32_t n_kmers, int32_t end_event, abea_index_pair_t* map) {
    if (n_kmers < 1 || !count || !map) return abea_fail(ABEA_EINVAL, "abea_expand_kmer_counts_to_map: bad argument");
    if (!expand_counts_to_map(count, n_kmers, end_event, map)) return abea_fail(ABEA_EINVAL, "a k-mer holds 255 or more events: use the walk");
    return ABEA_OK;
}

extern "C" int abea_expand_walk_codes_to_map(const uint32_t* codes, int32_t n_steps, int32_t last_kmer, int32_t end_event,
                                             abea_index_pair_t* map) {
    if (n_steps < 1 || !codes || !map || last_kmer < 0) return abea_fail(ABEA_EINVAL, "abea_expand_walk_codes_to_map: bad argument");
    expand_codes_to_map(codes, n_steps, last_kmer, end_event, map);
    return ABEA_OK;
}

/* host-only entry over the same routine, so that the expansion can be unit-tested without a GPU */
extern "C" int abea_expand_walk_codes(const uint32_t* codes, int32_t n_steps, int32_t last_kmer, int32_t end_event, abea_pair_t* out) {
    if (n_steps < 0 || (n_steps && (!codes || !out))) return abea_fail(ABEA_EINVAL, "abea_expand_walk_codes: bad argument");
    expand_codes(codes, n_steps, last_kmer, end_event, out);
    return ABEA_OK;
}

struct host_opts {
    size_t chunk_events;
    int32_t chunk_reads_min, chunk_reads_max;
    int n_slots;
    bool device_pairs;
    bool sdma_d2h;             /* experiment: return the result block with hipMemcpyAsync instead of the copy-out kernel */
    int flatten_prefetch;      /* bytes the flatten loop prefetches ahead of its loads (0 = none) */
    int flatten_hint;          /* 0 = prefetchnta, 1 = prefetcht0, 2 = prefetcht2 */
    bool ramp_order;           /* launch order: an ascending ramp of long reads first, then longest-first (order_for_launch) */
};

static host_opts read_opts() {
    host_opts o;
    /* >= 1024 reads and >= 24 M events per chunk on 8 slots (round 5, 16 hardware queues, spread workers, configs[2] in one process:
     * 344 ms against 355-357 with 2048 reads / 48 M events, 384 with 4096 / 96 M, 392 with 16 slots: profiles/r05/k_chunk_slot_sweep.txt) */
    o.chunk_events = (size_t)24 << 20; o.chunk_reads_min = 1024; o.chunk_reads_max = 16384; o.n_slots = 8; o.device_pairs = false;
    if (const char* e = getenv("ABEA_HOST_CHUNK_EVENTS")) o.chunk_events = std::max<size_t>(1, strtoull(e, nullptr, 10));
    if (const char* e = getenv("ABEA_HOST_CHUNK_READS")) o.chunk_reads_min = std::max(1, atoi(e));
    if (const char* e = getenv("ABEA_HOST_CHUNK_READS_MAX")) o.chunk_reads_max = std::max(1, atoi(e));
    if (const char* e = getenv("ABEA_HOST_SLOTS")) o.n_slots = std::min(ABEA_MAX_SLOTS, std::max(1, atoi(e)));
    if (const char* e = getenv("ABEA_HOST_PAIRS")) o.device_pairs = strcmp(e, "device") == 0;
    o.sdma_d2h = getenv("ABEA_HOST_SDMA_D2H") != nullptr;
    o.flatten_prefetch = 1536; o.flatten_hint = 1;      /* prefetcht0 1.5 KB ahead: +5 % on the MI355X host (2 x EPYC), tools/host_sweep.py */
    o.ramp_order = true;
    if (const char* e = getenv("ABEA_HOST_ORDER")) o.ramp_order = strcmp(e, "lpt") != 0;
    if (const char* e = getenv("ABEA_HOST_FLATTEN_PREFETCH")) o.flatten_prefetch = std::max(0, std::min(1 << 16, atoi(e)));
    if (const char* e = getenv("ABEA_HOST_FLATTEN_HINT")) o.flatten_hint = std::max(0, std::min(2, atoi(e)));
    o.chunk_reads_max = std::max(o.chunk_reads_max, o.chunk_reads_min);
    return o;
}

/* ------------------------------------------------------------------ chunk carving (pure host logic) */
struct chunk_span { size_t begin, end; size_t events; bool whole_arena; };

/* device + pinned bytes one read adds to a chunk besides scratch_bytes() */
static size_t chunk_io_bytes(const plan_read& r, bool pairs_on_device, bool scaling) {
    size_t b = align_up((size_t)r.L + 1, 16) + 4 + sizeof(abea_read_diag) + 8;
    if (pairs_on_device) b += ((size_t)r.E + (size_t)r.L) * sizeof(abea_pair_t);
    /* map (device scratch) + per-read scalars */
    if (scaling) b += (size_t)r.K * (sizeof(abea_index_pair_t) + 1) + sizeof(abea_scalings_t) + 8 + 8 + 4 + 4 + 4;   /* + one count byte per k-mer */
    return b + 64;
}

/* Cut the reads `order` (positions into `reads`, longest first) into chunks: a chunk closes once it holds at least
 * chunk_reads_min reads AND chunk_events events (the first two chunks a quarter / half of that, so the GPU starts
 * early), at chunk_reads_max reads, or when the next read would not fit the slot's share of the arena; a read that
 * does not fit a share on its own gets the whole arena, alone. */
static std::vector<chunk_span> carve_chunks(const std::vector<plan_read>& reads, const std::vector<int32_t>& order,
                                            const host_opts& opt, size_t slot_arena, bool pairs_on_device, bool scaling) {
    std::vector<chunk_span> out;
    size_t pos = 0;
    while (pos < order.size()) {
        const int ramp = out.size() == 0 ? 4 : out.size() == 1 ? 2 : 1;
        const size_t want_ev = opt.chunk_events / (size_t)ramp;
        const int32_t want_reads = std::max(1, opt.chunk_reads_min / ramp);
        size_t bytes = 65536, ev = 0, end = pos;
        bool whole_arena = false;
        while (end < order.size()) {
            const plan_read& r = reads[(size_t)order[end]];
            const size_t need = scratch_bytes(r) + chunk_io_bytes(r, pairs_on_device, scaling);
            if (bytes + need > slot_arena) {
                if (end > pos) break;
                whole_arena = true;                      /* an over-long read: give it the whole arena, alone */
            }
            bytes += need; ev += (size_t)r.E; ++end;
            const int32_t cnt = (int32_t)(end - pos);
            if (whole_arena || (cnt >= want_reads && ev >= want_ev) || cnt >= opt.chunk_reads_max) break;
        }
        out.push_back(chunk_span{pos, end, ev, whole_arena});
        pos = end;
    }
    return out;
}

/* longest first (band count), ties in caller order.  `order` arrives in ascending caller order; a stable LSD radix sort on the
 * complemented band count (four 7-bit digits cover ABEA_MAX_BANDS = 2^27) keeps it for equal keys.  0.6 ms for 100 k reads
 * where std::sort of packed keys took 7 ms of the caller's thread before the first chunk could be cut (round 5). */
static void order_longest_first(const std::vector<plan_read>& reads, std::vector<int32_t>& order) {
    const size_t n = order.size();
    if (n < 2) return;
    std::vector<uint32_t> key(n), key2(n);
    std::vector<int32_t> ord2(n);
    for (size_t t = 0; t < n; ++t)
        key[t] = (uint32_t)(ABEA_MAX_BANDS - std::min<int64_t>(ABEA_MAX_BANDS, std::max<int64_t>(0, reads[(size_t)order[t]].n_bands)));
    uint32_t* k_in = key.data(); uint32_t* k_out = key2.data();
    int32_t* o_in = order.data(); int32_t* o_out = ord2.data();
    for (int pass = 0; pass < 4; ++pass) {                  /* 4 x 7 bits = 28 bits >= log2(2^27 + 1) */
        const int sh = pass * 7;
        size_t cnt[129] = {0};
        for (size_t t = 0; t < n; ++t) ++cnt[((k_in[t] >> sh) & 127u) + 1];
        for (int b = 0; b < 128; ++b) cnt[b + 1] += cnt[b];
        for (size_t t = 0; t < n; ++t) {
            const size_t at = cnt[(k_in[t] >> sh) & 127u]++;
            k_out[at] = k_in[t]; o_out[at] = o_in[t];
        }
        std::swap(k_in, k_out); std::swap(o_in, o_out);
    }
    /* four passes: the result is back in `order` */
}

/* The order in which the reads are launched.  Longest-processing-time-first is what the END of a batch wants (the last reads
 * to start must be short), but it is the worst start: a wavefront holds a read for its whole length, the GPU is at full rate
 * only with ~4096 reads resident, and the 4096 longest reads of a 1-50 kb mix are 390 M events — 37 ms of flatten during which
 * the device runs at half occupancy on average.  Any read length keeps the wave slots full once they ARE full (the host
 * supplies events 1.5x faster than the kernel consumes them), so the batch starts with an ASCENDING RAMP: every other read of
 * those above ABEA_RAMP_BANDS bands, shortest first — 4096 of the ramp's first reads are flattened in ~5 ms, and the read
 * length then climbs to the longest while the slots stay full — followed by all the other reads longest first (the LPT tail is
 * untouched).  Worth 15-18 ms per 100 k reads (367 -> 350 ms); batches with fewer than ABEA_RAMP_MIN_READS long
 * reads keep the plain longest-first order.  Results do not depend on the order.  ABEA_HOST_ORDER=lpt restores longest-first. */
static const int64_t ABEA_RAMP_BANDS = 18000;         /* ~12 k events: 4096 such reads are 50 M events */
static const size_t ABEA_RAMP_MIN_READS = 8192;
static void order_for_launch(const std::vector<plan_read>& reads, std::vector<int32_t>& order, const host_opts& opt) {
    order_longest_first(reads, order);
    if (!opt.ramp_order) return;
    size_t p = 0;
    while (p < order.size() && reads[(size_t)order[p]].n_bands >= ABEA_RAMP_BANDS) ++p;
    if (p < ABEA_RAMP_MIN_READS) return;
    std::vector<int32_t> out; out.reserve(order.size());
    for (size_t t = (p - 1) | 1; ; t -= 2) {             /* odd ranks below p, ascending length */
        if (t < p) out.push_back(order[t]);
        if (t < 2) break;
    }
    for (size_t t = 0; t < order.size(); ++t) if (t >= p || (t & 1) == 0) out.push_back(order[t]);   /* the rest, longest first */
    order.swap(out);
}

/* The chunk plan abea_align_batch_host would use for these reads on an arena of `arena_bytes` (pairs returned through
 * the walk codes, no fused scaling), host-only: chunk_of[i] = chunk of read i, -1 for reads the align_single guards
 * skip; chunks are numbered in launch order.  Exposed so that the planning logic is testable without a GPU. */
extern "C" int abea_host_plan_chunks(const int32_t* read_len, const int32_t* n_events, int32_t n_reads, uint32_t kmer_size,
                                     uint64_t arena_bytes, int32_t* chunk_of, int32_t* n_chunks) {
    if (n_reads < 0 || (n_reads && (!read_len || !n_events || !chunk_of)) || !n_chunks || kmer_size < 1)
        return abea_fail(ABEA_EINVAL, "abea_host_plan_chunks: bad argument");
    const host_opts opt = read_opts();
    std::vector<plan_read> reads((size_t)n_reads);
    std::vector<int32_t> order;
    for (int32_t i = 0; i < n_reads; ++i) {
        reads[(size_t)i] = make_plan(i, read_len[i], n_events[i], kmer_size);
        chunk_of[i] = -1;
        if (reads[(size_t)i].run) {
            if (scratch_bytes(reads[(size_t)i]) + chunk_io_bytes(reads[(size_t)i], false, false) + 65536 > arena_bytes)
                return abea_fail(ABEA_ENOMEM, "read %d (L=%d, E=%d) needs more scratch than the %llu-byte arena", i, read_len[i],
                                 n_events[i], (unsigned long long)arena_bytes);
            order.push_back(i);
        }
    }
    order_for_launch(reads, order, opt);
    const size_t slot_arena = (size_t)arena_bytes / (size_t)opt.n_slots / 4096 * 4096;
    const std::vector<chunk_span> chunks = carve_chunks(reads, order, opt, slot_arena, false, false);
    for (size_t c = 0; c < chunks.size(); ++c)
        for (size_t t = chunks[c].begin; t < chunks[c].end; ++t) chunk_of[order[t]] = (int32_t)c;
    *n_chunks = (int32_t)chunks.size();
    return ABEA_OK;
}

/* ------------------------------------------------------------------ flatten: event_t.mean -> float array */
/* 4 of event_t's 24 bytes are wanted, but every cache line of the table crosses the memory bus: the loop is bound by how many
 * line fills one core keeps in flight (16 GB/s per thread measured in situ on the MI355X host against 23-32 GB/s for one thread
 * alone; 14 threads is all a 16-CPU quota leaves).  Eight events (three lines) per iteration with a software prefetch per line
 * `pf` bytes ahead of the loads; non-temporal 16-byte stores (the staging block is written once and read by the DMA engine only;
 * dst is 16-byte aligned: evm_off is a multiple of 4 floats). */
template <int HINT>
static inline void flatten_means_t(const abea_event_t* evs, int32_t E, float* dst, int pf) {
    const char* base = reinterpret_cast<const char*>(evs);
    const char* const last = base + (size_t)E * sizeof(abea_event_t) - 1;
    int32_t e = 0;
    if (pf > 0) {
        for (; e + 8 <= E; e += 8) {
            const char* p = base + (size_t)e * sizeof(abea_event_t) + pf;           /* 192 B per iteration = three lines */
            if (p + 191 <= last) {
                __builtin_prefetch(p, 0, HINT); __builtin_prefetch(p + 64, 0, HINT); __builtin_prefetch(p + 128, 0, HINT);
            }
            _mm_stream_ps(dst + e, _mm_set_ps(evs[e + 3].mean, evs[e + 2].mean, evs[e + 1].mean, evs[e].mean));
            _mm_stream_ps(dst + e + 4, _mm_set_ps(evs[e + 7].mean, evs[e + 6].mean, evs[e + 5].mean, evs[e + 4].mean));
        }
    }
    for (; e + 4 <= E; e += 4)
        _mm_stream_ps(dst + e, _mm_set_ps(evs[e + 3].mean, evs[e + 2].mean, evs[e + 1].mean, evs[e].mean));
    for (; e < E; ++e) dst[e] = evs[e].mean;
}
static inline void flatten_means(const abea_event_t* evs, int32_t E, float* dst, int pf, int hint) {
    /* __builtin_prefetch locality: 0 = prefetchnta, 3 = prefetcht0, 1 = prefetcht2 */
    if (hint == 1) flatten_means_t<3>(evs, E, dst, pf);
    else if (hint == 2) flatten_means_t<1>(evs, E, dst, pf);
    else flatten_means_t<0>(evs, E, dst, pf);
}

/* host-only entry over the same loop (unit tests, tools/probe/flatten_ab.cpp): means[e] = events[e].mean */
extern "C" int abea_flatten_event_means(const abea_event_t* events, int32_t n_events, float* means, int32_t prefetch_bytes, int32_t hint) {
    if (n_events < 0 || (n_events && (!events || !means)) || (reinterpret_cast<uintptr_t>(means) & 15u))
        return abea_fail(ABEA_EINVAL, "abea_flatten_event_means: bad argument (means must be 16-byte aligned)");
    flatten_means(events, n_events, means, std::max(0, prefetch_bytes), hint);
    _mm_sfence();
    return ABEA_OK;
}

/* ------------------------------------------------------------------ the pipeline on one device */
struct host_run_state {
    double t_origin = 0; bool trace = false;
    void log(const char* what, int chunk, int m = 0, size_t ev = 0) const {
        if (trace) fprintf(stderr, "[abea host dev %d] %8.2f ms  chunk %2d  %-14s reads %d events %zu\n", c->device, abea_now_ms() - t_origin, chunk, what, m, ev);
    }
    abea_ctx* c;
    abea_host_lane* lane;
    const abea_host_batch* H;
    hipEvent_t origin = nullptr;          /* GPU clock origin of the call (recorded on the idle context stream) */
    std::vector<std::pair<float, float>> spans;   /* kernel span of every retired chunk on that clock */
    std::atomic<bool> oom{false};         /* ABEA_HB_MALLOC_MAPS: a map could not be allocated */
    host_opts opt;
    bool want_pairs, scaling, device_pairs;
    abea_stats st;
    std::vector<plan_read> reads;         /* indexed by position in `mine` */
};

/* stage A of a device-pairs chunk: the kernels are done, the total pair count is known -> copy the compacted lists */
static int slot_stage(host_run_state& S, abea_host_slot& sl, bool block) {
    if (!sl.busy || !sl.device_pairs || sl.staged) return ABEA_OK;
    if (!block && hipEventQuery(sl.kdone) != hipSuccess) return ABEA_OK;
    const double t0 = abea_now_ms();
    HIP_TRY(hipEventSynchronize(sl.kdone));
    S.st.wait_ms += abea_now_ms() - t0;
    const unsigned long long total = *(const unsigned long long*)(sl.dn + sl.o_cursor);
    if (total > sl.pair_cap) return abea_fail(ABEA_EHIP, "internal: %llu compacted pairs exceed the capacity %zu", total, sl.pair_cap);
    if (total)
        HIP_TRY(hipMemcpyAsync(sl.dn + sl.o_pairs, sl.d_pairs, (size_t)total * sizeof(abea_pair_t), hipMemcpyDeviceToHost, sl.post));
    HIP_TRY(hipEventRecord(sl.done, sl.post));
    S.st.d2h_bytes += total * sizeof(abea_pair_t);
    sl.staged = true;
    return ABEA_OK;
}

/* Finish the chunk in flight in `sl`: wait for its results, then un-flatten into the caller-owned per-read buffers
 * (the role of f5c.cu:1003-1030). */
static int slot_retire(host_run_state& S, abea_host_slot& sl) {
    if (!sl.busy) return ABEA_OK;
    int rc = slot_stage(S, sl, true);
    if (rc) return rc;
    double t0 = abea_now_ms();
    S.log("wait", sl.chunk_no);
    HIP_TRY(hipEventSynchronize(sl.done));
    S.st.wait_ms += abea_now_ms() - t0;
    S.log("unflatten", sl.chunk_no);
    const abea_host_batch* H = S.H;
    const abea_read_desc* descs = (const abea_read_desc*)sl.up;
    const int32_t* npairs = (const int32_t*)(sl.dn + sl.o_np);
    const abea_read_diag* diag = (const abea_read_diag*)(sl.dn + sl.o_diag);
    const uint32_t* codes = (const uint32_t*)(sl.dn + sl.o_codes);
    const int64_t* poff = (const int64_t*)(sl.dn + sl.o_poff);
    const abea_pair_t* pairs = (const abea_pair_t*)(sl.dn + sl.o_pairs);
    const abea_scalings_t* sc = (const abea_scalings_t*)(sl.dn + sl.o_sc);
    const double* epb = (const double*)(sl.dn + sl.o_epb);
    const int32_t* flag = (const int32_t*)(sl.dn + sl.o_flag);
    const int32_t* nal = (const int32_t*)(sl.dn + sl.o_nal);
    const double* var64 = (const double*)(sl.dn + sl.o_var);
    const uint8_t* kcnt = (const uint8_t*)(sl.dn + sl.o_cnt);
    t0 = abea_now_ms();
    const bool want_pairs = S.want_pairs, dev_pairs = sl.device_pairs, scaling = sl.scaling;
    S.lane->pool->run(sl.m, 1, [&](int64_t lo, int64_t hi) {
        for (int64_t j = lo; j < hi; ++j) {
            const int32_t i = sl.rd[(size_t)j];
            const int32_t np = npairs[j];
            H->n_pairs[i] = np;
            if (H->diag) H->diag[i] = diag[j];
            if (want_pairs && np > 0) {
                if (dev_pairs) memcpy(H->pairs[i], pairs + poff[j], (size_t)np * sizeof(abea_pair_t));
                else expand_codes(codes + descs[j].code_off, np, descs[j].n_kmers - 1, diag[j].best_event, H->pairs[i]);
            }
            if (scaling) {
                /* the map comes down as one byte per k-mer (0.5 B per event on the wire) and is rebuilt with two adds per entry;
                 * a read with a k-mer of 255+ events falls back to the walk (0.4 B per event, which comes down anyway) */
                abea_index_pair_t* map = H->base_to_event_map[i];
                if (H->flags & ABEA_HB_MALLOC_MAPS) {      /* scaling_single mallocs the map of an aligned read (f5c.c:746), NULL otherwise (f5c.c:787) */
                    map = np > 0 ? (abea_index_pair_t*)malloc(sizeof(abea_index_pair_t) * (size_t)descs[j].n_kmers) : nullptr;
                    if (np > 0 && !map) S.oom.store(true);
                    const_cast<abea_index_pair_t**>(H->base_to_event_map)[i] = map;
                }
                if (np > 0 && map &&
                    !expand_counts_to_map(kcnt + descs[j].kmer_off, descs[j].n_kmers, diag[j].best_event, map))
                    expand_codes_to_map(codes + descs[j].code_off, np, descs[j].n_kmers - 1, diag[j].best_event, map);
                if (H->scalings_out) {
                    abea_scalings_t o = sc[j];
                    abea_apply_log_var(o, var64[j]);                             /* align.c:758-760 (CACHED_LOG): double log, glibc's */
                    H->scalings_out[i] = o;
                }
                if (H->events_per_base) H->events_per_base[i] = epb[j];
                if (H->read_stat_flag) H->read_stat_flag[i] = flag[j];
                if (H->n_event_alignment) H->n_event_alignment[i] = nal[j];
            }
        }
    });
    S.st.unflatten_ms += abea_now_ms() - t0;
    if (S.oom.load()) return abea_fail(ABEA_ENOMEM, "malloc of a base_to_event_map failed");
    S.log("retired", sl.chunk_no);
    float ms = 0;
    if (S.origin) {                       /* the chunk's kernels on the GPU's clock: start of align-pre, start and end of the alignment kernel */
        float a = 0, b = 0, e = 0;
        if (hipEventElapsedTime(&a, S.origin, sl.k0) == hipSuccess && hipEventElapsedTime(&b, S.origin, sl.k1) == hipSuccess &&
            hipEventElapsedTime(&e, S.origin, sl.k2) == hipSuccess) {
            S.spans.emplace_back(a, e);
            if (S.trace) fprintf(stderr, "[abea host dev %d] gpu chunk %2d  pre %9.3f  align %9.3f .. %9.3f ms  reads %d\n", S.c->device, sl.chunk_no, a, b, e, sl.m);
        }
    }
    HIP_TRY(hipEventElapsedTime(&ms, sl.k0, sl.k1)); S.st.pre_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, sl.k1, sl.k2)); S.st.fill_ms += ms;
    for (int32_t j = 0; j < sl.m; ++j) S.st.sum_pairs += npairs[j];
    sl.busy = false;
    return ABEA_OK;
}

/* on every exit, error or not, nothing of this call may stay in flight: the slots' bookkeeping and the pinned /
 * caller buffers they point at belong to this batch only */
struct slot_guard {
    abea_ctx* c; const abea_host_lane* lane;
    ~slot_guard() {
        for (int q = lane->first_slot; q < lane->first_slot + lane->n_slots && q < (int)c->slots.size(); ++q) {
            abea_host_slot* s = c->slots[(size_t)q];
            if (s && s->busy) { hipStreamSynchronize(s->stream); if (s->stream_hi) hipStreamSynchronize(s->stream_hi); s->busy = false; }
        }
    }
};

/* the lanes of a DEVICE context, built on first use: the full lane and n_lanes equal shares.  Slot objects (stream,
 * events, pinned staging) are created on demand by host_run; a slot index belongs to exactly one share and the full lane
 * is only used while no share is (the top-level context's bookkeeping guarantees it). */
static void ensure_lanes(abea_ctx* c, const host_thread_plan& pl, int n_lanes, int n_slots, bool want_shares) {
    abea_host_async* a = async_of(c);
    if (!a->full.pool || a->full.pool->key != pl.key()) {
        delete a->full.pool;
        a->full.pool = new abea_host_pool(pl.threads, pl.cpus, pl.spread);
        a->full.pool->key = pl.key();
    }
    a->full.first_slot = 0; a->full.n_slots = n_slots; a->full.arena_off = 0; a->full.arena_bytes = c->arena_bytes;
    /* the share lanes (and their worker pools) exist only once a batch has been SUBMITTED: a context that is only ever
     * called synchronously keeps one pool per device (round-3 advisor finding: 30 threads per device instead of 16) */
    if (!want_shares) return;
    const int per_threads = std::max(1, pl.threads / std::max(1, n_lanes));
    if ((int)a->lanes.size() != n_lanes || (n_lanes && a->lanes[0].pool->threads() != per_threads)) {
        for (abea_host_lane& l : a->lanes) delete l.pool;
        a->lanes.assign((size_t)n_lanes, abea_host_lane());
        const int per_slots = std::max(1, n_slots / std::max(1, n_lanes));
        const size_t per_arena = c->arena_bytes / (size_t)std::max(1, n_lanes) / 4096 * 4096;
        for (int l = 0; l < n_lanes; ++l) {
            a->lanes[(size_t)l].first_slot = l * per_slots; a->lanes[(size_t)l].n_slots = per_slots;
            a->lanes[(size_t)l].arena_off = (size_t)l * per_arena; a->lanes[(size_t)l].arena_bytes = per_arena;
            a->lanes[(size_t)l].pool = new abea_host_pool(per_threads, pl.cpus, pl.spread);
        }
    }
}

static hipStream_t sl0_stream(abea_ctx* c, int slot0) { return c->slots[(size_t)slot0]->stream; }

static int host_run(abea_ctx* c, const abea_host_batch* H, const int32_t* mine, int32_t n_mine, abea_host_lane& lane,
                    abea_stats* st_out) {
    const double t_start = abea_now_ms();
    HIP_TRY(hipSetDevice(c->device));          /* the caller's thread changes per batch (f5c.cu:692-694) */
    host_run_state S;
    S.c = c; S.lane = &lane; S.H = H; S.opt = read_opts();
    S.t_origin = t_start; S.trace = getenv("ABEA_HOST_TRACE") != nullptr;
    memset(&S.st, 0, sizeof S.st);
    S.st.arena_bytes = lane.arena_bytes;
    S.st.n_devices = 1;
    S.scaling = H->base_to_event_map != nullptr;
    S.want_pairs = H->pairs != nullptr;
    S.device_pairs = S.want_pairs && S.opt.device_pairs && !S.scaling;
    const bool scaling = S.scaling;
    const bool pairs_on_device = S.device_pairs;                  /* fused scaling_single builds its map from the walk, not from pair lists */

    /* ---- worker pool and slots (persistent across calls) ---- */
    S.st.host_threads = lane.pool->threads();
    const int n_slots = lane.n_slots, slot0 = lane.first_slot;
    {
        std::lock_guard<std::mutex> lk(c->slots_mu);             /* sized once: lanes of one device index it concurrently */
        if (c->slots.size() < (size_t)ABEA_MAX_SLOTS) c->slots.resize((size_t)ABEA_MAX_SLOTS, nullptr);
    }
    for (int q = slot0; q < slot0 + n_slots; ++q) {
        if (!c->slots[(size_t)q]) {
            abea_host_slot* s = nullptr;
            const int rc = slot_create(&s);
            c->slots[(size_t)q] = s;
            if (rc) return rc;
        }
        c->slots[(size_t)q]->busy = false;                       /* nothing survives a call (slot_guard) */
    }
    slot_guard guard{c, &lane};
    struct origin_event {                                        /* per call: several lanes of one context may run at once */
        hipEvent_t e = nullptr;
        ~origin_event() { if (e) hipEventDestroy(e); }
    } origin;
    if (hipEventCreate(&origin.e) == hipSuccess && hipEventRecord(origin.e, sl0_stream(c, slot0)) == hipSuccess) {
        S.origin = origin.e;
        if (S.trace && hipEventSynchronize(origin.e) == hipSuccess) S.t_origin = abea_now_ms();   /* host and GPU clocks share (to ~20 us) this origin */
    }
    uint8_t* const lane_arena = c->arena + lane.arena_off;
    auto slot_at = [&](int q) -> abea_host_slot& { return *c->slots[(size_t)(slot0 + q)]; };

    /* ---- guards (align_single, f5c.c:811-830) and ordering ---- */
    S.reads.resize((size_t)n_mine);
    std::vector<int32_t> order; order.reserve((size_t)n_mine);
    for (int32_t q = 0; q < n_mine; ++q) {
        const int32_t i = mine ? mine[q] : q;
        const bool good = (!H->n_samples || H->n_samples[i] > 0) && H->read[i] && H->events[i] &&
                          (!S.want_pairs || H->pairs[i]) && H->read_len[i] > 0 && H->n_events[i] > 0 &&
                          H->n_events[i] < (uint64_t)INT32_MAX;
        /* bad read (nsample == 0): n_pairs = 0 (f5c.c:826-828) */
        plan_read r = make_plan(i, good ? H->read_len[i] : 0, good ? (int32_t)H->n_events[i] : 0, c->k);
        if (r.run && r.n_bands > ABEA_MAX_BANDS)
            return abea_fail(ABEA_EINVAL, "read %d has %lld bands; the limit is %lld", i, (long long)r.n_bands, (long long)ABEA_MAX_BANDS);
        S.reads[(size_t)q] = r;
        if (r.run) { order.push_back(q); ++S.st.n_reads_gpu; }
        else {
            ++S.st.n_reads_skipped;
            H->n_pairs[i] = 0;
            if (H->diag) {
                abea_read_diag dg; memset(&dg, 0, sizeof dg);
                dg.max_score = -__builtin_inff(); dg.flags = ABEA_RF_SKIPPED;
                H->diag[i] = dg;
            }
            if (scaling) {                                        /* f5c.c:786-794: could not align */
                if (H->flags & ABEA_HB_MALLOC_MAPS) const_cast<abea_index_pair_t**>(H->base_to_event_map)[i] = nullptr;
                if (H->scalings_out) H->scalings_out[i] = H->scalings[i];
                if (H->events_per_base) H->events_per_base[i] = 0.0;
                if (H->read_stat_flag) H->read_stat_flag[i] |= ABEA_FAILED_ALIGNMENT;
                if (H->n_event_alignment) H->n_event_alignment[i] = 0;
            }
        }
    }
    order_for_launch(S.reads, order, S.opt);
    auto io_bytes = [&](const plan_read& r) { return chunk_io_bytes(r, pairs_on_device, scaling); };
    /* check every read against the arena before anything is launched */
    for (int32_t q : order) {
        const plan_read& r = S.reads[(size_t)q];
        if (scratch_bytes(r) + io_bytes(r) + 65536 > lane.arena_bytes)
            return abea_fail(ABEA_ENOMEM, "read %d (L=%d, E=%d) needs more scratch than the %zu-byte arena", r.idx, r.L, r.E,
                             lane.arena_bytes);
    }
    const size_t slot_arena = lane.arena_bytes / (size_t)n_slots / 4096 * 4096;
    const int min_rescale = H->min_num_events_to_rescale > 0 ? H->min_num_events_to_rescale : 200;

    const std::vector<chunk_span> chunks = carve_chunks(S.reads, order, S.opt, slot_arena, pairs_on_device, scaling);
    S.st.setup_ms = abea_now_ms() - t_start;
    int turn = 0;
    for (int chunk_no = 0; chunk_no < (int)chunks.size(); ++chunk_no) {
        const size_t pos = chunks[(size_t)chunk_no].begin, end = chunks[(size_t)chunk_no].end, ev = chunks[(size_t)chunk_no].events;
        const bool whole_arena = chunks[(size_t)chunk_no].whole_arena;
        const int32_t m = (int32_t)(end - pos);
        abea_host_slot& sl = slot_at(whole_arena ? 0 : turn % n_slots);
        int rc;
        if (whole_arena) { for (int q = 0; q < n_slots; ++q) if ((rc = slot_retire(S, slot_at(q)))) return rc; }
        else if ((rc = slot_retire(S, sl))) return rc;
        uint8_t* arena = whole_arena ? lane_arena : lane_arena + (size_t)(turn % n_slots) * slot_arena;

        /* ---- plan the chunk ---- */
        double t0 = abea_now_ms();
        S.log("plan", chunk_no, m, ev);
        sl.chunk_no = chunk_no;
        sl.m = m; sl.scaling = scaling; sl.device_pairs = S.device_pairs; sl.staged = false;
        sl.rd.resize((size_t)m);
        size_t n_read = 0, n_pair = 0, n_kmer = 0;
        for (int32_t j = 0; j < m; ++j) {
            const plan_read& r = S.reads[(size_t)order[pos + (size_t)j]];
            n_read += align_up((size_t)r.L + 1, 16);
            n_pair += (size_t)r.E + (size_t)r.L;
            n_kmer += (size_t)r.K;
        }
        size_t n_evm_max = 0;
        for (int32_t j = 0; j < m; ++j) n_evm_max += align_up((size_t)S.reads[(size_t)order[pos + (size_t)j]].E + 64, 4);
        /* `up` = [desc][reads][evm], mirrored at the start of the chunk's arena share: one H2D copy */
        const size_t u_desc = 0;
        const size_t u_reads = align_up(u_desc + (size_t)m * sizeof(abea_read_desc), 256);
        const size_t u_evm = align_up(u_reads + n_read, 256);
        const size_t u_end = align_up(u_evm + n_evm_max * 4 + 512, 256);
        if ((rc = ensure_pinned((void**)&sl.up, &sl.up_cap, u_end))) return rc;
        abea_read_desc* descs = (abea_read_desc*)(sl.up + u_desc);
        sub_layout lay;
        /* the serial part of the plan: every read's offsets into the chunk's arrays, into a small cache-resident table; the
         * descriptors themselves (112 B each into the pinned staging block: two write misses per read) are written by the
         * parallel flatten loop below */
        sl.offs.resize((size_t)m);
        {
            size_t ro = 0, po = 0, ko = 0;
            for (int32_t j = 0; j < m; ++j) {
                const plan_read& r = S.reads[(size_t)order[pos + (size_t)j]];
                sl.rd[(size_t)j] = r.idx;
                abea_host_slot::read_offs& f = sl.offs[(size_t)j];
                f.o = plan_advance(r, lay, S.st);
                f.src = order[pos + (size_t)j];
                f.read_off = (int64_t)ro; ro += align_up((size_t)r.L + 1, 16);
                f.pair_off = (int64_t)po; po += (size_t)r.E + (size_t)r.L;
                f.kmer_off = (int64_t)ko; ko += (size_t)r.K;
            }
        }
        /* `dn` = [npairs][diag][codes | poff, cursor][scaling outputs] mirrors the arena block behind the scratch */
        size_t o = 0;
        sl.o_np = o;      o = align_up(o + (size_t)m * 4, 256);
        sl.o_diag = o;    o = align_up(o + (size_t)m * sizeof(abea_read_diag), 256);
        sl.o_codes = o;   if (!S.device_pairs && (S.want_pairs || scaling)) o = align_up(o + lay.n_code * 4, 256);
        sl.o_poff = o;    if (S.device_pairs) o = align_up(o + (size_t)m * 8, 256);
        sl.o_cursor = o;  if (S.device_pairs) o = align_up(o + 8, 256);
        sl.o_sc = o;      if (scaling) o = align_up(o + (size_t)m * sizeof(abea_scalings_t), 256);
        sl.o_epb = o;     if (scaling) o = align_up(o + (size_t)m * 8, 256);
        sl.o_flag = o;    if (scaling) o = align_up(o + (size_t)m * 4, 256);
        sl.o_nal = o;     if (scaling) o = align_up(o + (size_t)m * 4, 256);
        sl.o_var = o;     if (scaling) o = align_up(o + (size_t)m * 8, 256);
        sl.o_cnt = o;     if (scaling) o = align_up(o + n_kmer, 256);   /* events per k-mer, one byte each: the map as it crosses PCIe */
        const size_t dn_copy = o;                         /* one D2H copy of [0, dn_copy) */
        sl.o_pairs = o;   if (S.device_pairs) o = align_up(o + n_pair * sizeof(abea_pair_t), 256);
        if ((rc = ensure_pinned((void**)&sl.dn, &sl.dn_cap, o))) return rc;

        /* ---- arena: [up block][kpar][trace][codes* ...] ; the dn block is contiguous on the device too ---- */
        uint8_t* p = arena;
        uint8_t* d_up = p;                                  p += u_end;
        abea_kpar_t* d_kpar = (abea_kpar_t*)p;              p += align_up(lay.n_kpar * sizeof(abea_kpar_t), 256);
        uint32_t* d_krank = (uint32_t*)p;                   p += align_up(lay.n_kpar * 4, 256);     /* k-mer ranks for phase 4 */
        uint4* d_trace = (uint4*)p;                         p += align_up(lay.n_trace * sizeof(uint4), 256);
        uint8_t* d_dn = p;                                  p += dn_copy;
        uint32_t* d_codes_scratch = nullptr;                /* codes stay on the device when nobody wants them */
        if (S.device_pairs) { d_codes_scratch = (uint32_t*)p; p += align_up(lay.n_code * 4, 256); }
        abea_pair_t* d_pairs = nullptr;
        if (pairs_on_device) { d_pairs = (abea_pair_t*)p; p += align_up(n_pair * sizeof(abea_pair_t), 256); }
        abea_index_pair_t* d_b2e = nullptr;
        if (scaling) {                                      /* device-only scratch of the fused scaling_single stage: the map */
            d_b2e = (abea_index_pair_t*)p;  p += align_up(n_kmer * sizeof(abea_index_pair_t), 256);
        }
        if ((size_t)(p - arena) > (whole_arena ? lane.arena_bytes : slot_arena))
            return abea_fail(ABEA_ENOMEM, "internal: chunk layout %zu exceeds its arena share %zu", (size_t)(p - arena),
                             whole_arena ? lane.arena_bytes : slot_arena);
        const abea_read_desc* d_desc = (const abea_read_desc*)(d_up + u_desc);
        const char* d_reads = (const char*)(d_up + u_reads);
        float* d_evm = (float*)(d_up + u_evm);
        int32_t* d_np = (int32_t*)(d_dn + sl.o_np);
        abea_read_diag* d_diag = (abea_read_diag*)(d_dn + sl.o_diag);
        uint32_t* d_codes = d_codes_scratch ? d_codes_scratch : (uint32_t*)(d_dn + sl.o_codes);
        int64_t* d_poff = (int64_t*)(d_dn + sl.o_poff);
        unsigned long long* d_cursor = (unsigned long long*)(d_dn + sl.o_cursor);
        sl.d_pairs = d_pairs; sl.pair_cap = n_pair;

        /* ---- flatten (the role of f5c.cu:744-802, which runs on one thread) ---- */
        char* h_reads = (char*)(sl.up + u_reads);
        float* h_evm = (float*)(sl.up + u_evm);
        S.log("flatten", chunk_no);
        const int pf = S.opt.flatten_prefetch, pf_hint = S.opt.flatten_hint;
        S.st.plan_ms += abea_now_ms() - t0;
        t0 = abea_now_ms();
        lane.pool->run(m, 1, [&](int64_t lo, int64_t hi) {
            for (int64_t j = lo; j < hi; ++j) {
                const int32_t i = sl.rd[(size_t)j];
                {
                    const abea_host_slot::read_offs& f = sl.offs[(size_t)j];
                    plan_read r = S.reads[(size_t)f.src];
                    r.idx = (int32_t)j;                     /* out_idx = position in the chunk */
                    abea_read_desc nd;
                    plan_desc_fill(nd, r, H->scalings[i], f.o);
                    nd.read_off = f.read_off; nd.pair_off = f.pair_off; nd.kmer_off = f.kmer_off;
                    plan_desc_consts(nd);                   /* align.c:207-216: four libm calls per read, off the caller's serial path */
                    descs[j] = nd;
                }
                const abea_read_desc& d = descs[j];
                const size_t L = (size_t)d.read_len;
                memcpy(h_reads + d.read_off, H->read[i], L);
                h_reads[d.read_off + (int64_t)L] = '\0';
                /* event means: 4 of event_t's 24 bytes (align.c:131 reads nothing else) */
                flatten_means(H->events[i], d.n_events, h_evm + d.evm_off, pf, pf_hint);
            }
            _mm_sfence();
        });
        S.st.flatten_ms += abea_now_ms() - t0;
        S.log("enqueue", chunk_no);

        /* ---- copy up, run, copy down ---- */
        HIP_TRY(hipMemcpyAsync(d_up, sl.up, u_evm + lay.n_evm * 4, hipMemcpyHostToDevice, sl.stream));
        S.st.h2d_bytes += u_evm + lay.n_evm * 4;
        if (S.device_pairs) HIP_TRY(hipMemsetAsync(d_cursor, 0, 8, sl.stream));
        if (scaling) {
            abea_scalings_t* h_sc = (abea_scalings_t*)(sl.dn + sl.o_sc);
            int32_t* h_flag = (int32_t*)(sl.dn + sl.o_flag);
            for (int32_t j = 0; j < m; ++j) {
                h_sc[j] = H->scalings[sl.rd[(size_t)j]];
                h_flag[j] = H->read_stat_flag ? H->read_stat_flag[sl.rd[(size_t)j]] : 0;
            }
            HIP_TRY(hipMemcpyAsync(d_dn + sl.o_sc, h_sc, (size_t)m * sizeof(abea_scalings_t), hipMemcpyHostToDevice, sl.stream));
            HIP_TRY(hipMemcpyAsync(d_dn + sl.o_flag, h_flag, (size_t)m * 4, hipMemcpyHostToDevice, sl.stream));
        }
        HIP_TRY(hipEventRecord(sl.k0, sl.stream));
        hipLaunchKernelGGL(abea_pre_kernel, dim3((unsigned)m), dim3(256), 0, sl.stream,
                           d_desc, d_reads, (const abea_event_t*)nullptr, c->d_model, (int)c->k, d_kpar, d_evm,
                           scaling ? d_krank : (uint32_t*)nullptr);
        HIP_TRY(hipEventRecord(sl.k1, sl.stream));
        /* scaling_single, when requested, is the last phase of the alignment kernel: the wavefront that aligned a read builds
         * its base_to_event_map and recalibrates its scalings (round 4; rounds 1-3 launched one / two kernels behind it, which
         * queued for wave slots behind the other chunks' alignment kernels and cost 4-15 ms per chunk) */
        abea_fused_scaling fs;
        memset(&fs, 0, sizeof fs);
        if (scaling) {
            fs.reads = d_reads; fs.model = c->d_model; fs.b2e = d_b2e;
            fs.sc_io = (abea_scalings_t*)(d_dn + sl.o_sc); fs.epb = (double*)(d_dn + sl.o_epb);
            fs.flag_io = (int32_t*)(d_dn + sl.o_flag); fs.nalign = (int32_t*)(d_dn + sl.o_nal);
            fs.kcnt = (uint8_t*)(d_dn + sl.o_cnt);
            fs.var_f64 = (double*)(d_dn + sl.o_var);
            fs.kmer_size = (int32_t)c->k; fs.min_rescale = min_rescale;
            fs.krank = d_krank; fs.mterms = c->d_mterms;
        }
        hipLaunchKernelGGL(abea_align_kernel, dim3((unsigned)m), dim3(64), 0, sl.stream,
                           d_desc, d_evm, d_kpar, d_trace, d_codes, d_pairs, d_np, d_diag,
                           S.device_pairs ? d_cursor : (unsigned long long*)nullptr, S.device_pairs ? d_poff : (int64_t*)nullptr, fs);
        HIP_TRY(hipEventRecord(sl.k2, sl.stream));
        /* only the small copy-out follows the kernel; it stays on the slot's stream unless ABEA_HOST_HI_STREAM=1 asks for the
         * high-priority queue (the measurements do not show a gain that outweighs one more busy queue) */
        hipStream_t post = sl.stream;
        if (sl.hi_always && (rc = slot_hi_stream(sl, &post))) return rc;
        sl.post = post;
        if (post != sl.stream) HIP_TRY(hipStreamWaitEvent(post, sl.k2, 0));
        /* the result block goes down by a kernel, not by an SDMA copy: a copy queued behind the alignment kernel would
         * hold its SDMA ring until that kernel ends and stall the next chunks' H2D copies (abea_copy_out_kernel) */
        if (S.opt.sdma_d2h) HIP_TRY(hipMemcpyAsync(sl.dn, d_dn, dn_copy, hipMemcpyDeviceToHost, post));
        else hipLaunchKernelGGL(abea_copy_out_kernel, dim3((unsigned)std::min<size_t>(512, (dn_copy / 16 + 255) / 256)), dim3(256), 0,
                                post, (const uint4*)d_dn, (uint4*)sl.dn, dn_copy / 16);
        HIP_TRY(hipGetLastError());
        S.st.d2h_bytes += dn_copy;
        if (S.device_pairs) HIP_TRY(hipEventRecord(sl.kdone, post));            /* the pair copy follows at stage A */
        else HIP_TRY(hipEventRecord(sl.done, post));
        sl.busy = true;
        S.log("enqueued", chunk_no);
        S.st.n_sub_batches += 1; S.st.fill_launches += 1;
        if (whole_arena) { if ((rc = slot_retire(S, sl))) return rc; }
        else ++turn;
        for (int q = 0; q < n_slots; ++q) if ((rc = slot_stage(S, slot_at(q), false))) return rc;
    }
    /* ---- drain, oldest chunk first ---- */
    for (int q = 0; q < n_slots; ++q) {
        int rc = slot_retire(S, slot_at((turn + q) % n_slots));
        if (rc) return rc;
    }
    S.st.host_ms = S.st.flatten_ms + S.st.unflatten_ms;
    S.st.gpu_busy_ms = interval_union_ms(S.spans);
    S.st.total_ms = abea_now_ms() - t_start;
    *st_out = S.st;
    return ABEA_OK;
}

/* ------------------------------------------------------------------ multi-device split */
/* Longest-processing-time-first on the band count E + K (f5c_amd/synth.py shard_batch is the same rule):
 * reads in descending weight go to the currently lightest device; ties by lowest device index.  `weight[i] <= 0`
 * reads (guard failures) come last and all land in whichever bin is lightest then: they cost nothing. */
extern "C" int abea_lpt_split(const int64_t* weight, int32_t n, int32_t n_bins, int32_t* bin_of) {
    if (!weight || !bin_of || n < 0 || n_bins < 1) return abea_fail(ABEA_EINVAL, "abea_lpt_split: bad argument");
    std::vector<int32_t> order((size_t)n);
    for (int32_t i = 0; i < n; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return weight[a] > weight[b]; });
    typedef std::pair<int64_t, int32_t> item;                 /* (load, bin): smallest load first, then lowest bin */
    std::priority_queue<item, std::vector<item>, std::greater<item>> heap;
    for (int32_t b = 0; b < n_bins; ++b) heap.push(item(0, b));
    for (int32_t i : order) {
        item t = heap.top(); heap.pop();
        bin_of[i] = t.second;
        heap.push(item(t.first + std::max<int64_t>(weight[i], 0), t.second));
    }
    return ABEA_OK;
}

static void stats_add(abea_stats& a, const abea_stats& b) {
    /* kernel / host times: the devices work at the same time, the batch waits for the slowest */
    a.pre_ms = std::max(a.pre_ms, b.pre_ms); a.fill_ms = std::max(a.fill_ms, b.fill_ms); a.trace_ms = std::max(a.trace_ms, b.trace_ms);
    a.host_ms = std::max(a.host_ms, b.host_ms); a.flatten_ms = std::max(a.flatten_ms, b.flatten_ms);
    a.unflatten_ms = std::max(a.unflatten_ms, b.unflatten_ms); a.wait_ms = std::max(a.wait_ms, b.wait_ms);
    a.plan_ms = std::max(a.plan_ms, b.plan_ms); a.setup_ms = std::max(a.setup_ms, b.setup_ms);
    a.gpu_busy_ms = std::max(a.gpu_busy_ms, b.gpu_busy_ms);
    a.n_reads_gpu += b.n_reads_gpu; a.n_reads_skipped += b.n_reads_skipped; a.n_sub_batches += b.n_sub_batches;
    a.sum_events += b.sum_events; a.sum_bands += b.sum_bands; a.sum_pairs += b.sum_pairs; a.fill_launches += b.fill_launches;
    a.arena_bytes += b.arena_bytes; a.bytes_ref += b.bytes_ref; a.bytes_min += b.bytes_min; a.bytes_moved += b.bytes_moved;
    a.h2d_bytes += b.h2d_bytes; a.d2h_bytes += b.d2h_bytes; a.host_threads += b.host_threads;
}

static int check_host_batch(const abea_host_batch* H) {
    if (!H) return abea_fail(ABEA_EINVAL, "null argument");
    if (H->n_reads < 0) return abea_fail(ABEA_EINVAL, "n_reads < 0");
    if (H->n_reads == 0) return ABEA_OK;
    if (!H->read || !H->read_len || !H->events || !H->n_events || !H->scalings || !H->n_pairs)
        return abea_fail(ABEA_EINVAL, "abea_align_batch_host: null array");
    if (!H->pairs && !H->base_to_event_map)
        return abea_fail(ABEA_EINVAL, "abea_align_batch_host: neither pairs nor base_to_event_map requested");
    return ABEA_OK;
}

/* temporarily move the calling thread onto the CPUs of its device's NUMA node (it takes part in the host loops and
 * first-touches what it allocates); restored on scope exit — the caller's thread is f5c's, not ours */
struct affinity_scope {
    cpu_set_t saved; bool active = false;
    explicit affinity_scope(const std::vector<int>& cpus) {
        if (cpus.empty() || sched_getaffinity(0, sizeof saved, &saved) != 0) return;
        active = true;
        bind_this_thread(cpus);
    }
    ~affinity_scope() { if (active) sched_setaffinity(0, sizeof saved, &saved); }
};

/* One host batch on lane `lane_no` (-1 = the full lane) of every device of the context: single device, or LPT split over
 * the children with one driver thread per device. */
static int run_host_batch_inner(abea_ctx* c, const abea_host_batch* H, int lane_no, int n_lanes, abea_stats* st_out);

static int run_host_batch(abea_ctx* c, const abea_host_batch* H, int lane_no, int n_lanes, abea_stats* st_out) {
    const bool own_maps = H->base_to_event_map && (H->flags & ABEA_HB_MALLOC_MAPS);
    abea_index_pair_t** maps = own_maps ? const_cast<abea_index_pair_t**>(H->base_to_event_map) : nullptr;
    if (own_maps) for (int32_t i = 0; i < H->n_reads; ++i) maps[i] = nullptr;
    const int rc = run_host_batch_inner(c, H, lane_no, n_lanes, st_out);
    if (rc && own_maps) for (int32_t i = 0; i < H->n_reads; ++i) { free(maps[i]); maps[i] = nullptr; }   /* nothing half-built is handed back */
    return rc;
}

static int run_host_batch_inner(abea_ctx* c, const abea_host_batch* H, int lane_no, int n_lanes, abea_stats* st_out) {
    const int32_t n = H->n_reads;
    const double t_start = abea_now_ms();
    const host_opts opt = read_opts();
    const std::vector<host_thread_plan> plan = context_thread_plan(c);
    auto lane_of = [&](abea_ctx* dev, const host_thread_plan& pl) -> abea_host_lane& {
        {
            std::lock_guard<std::mutex> lk(dev->slots_mu);
            ensure_lanes(dev, pl, n_lanes, opt.n_slots, lane_no >= 0);
        }
        abea_host_async* a = async_of(dev);
        return lane_no < 0 ? a->full : a->lanes[(size_t)lane_no];
    };
    if (c->children.empty()) {
        affinity_scope bound(plan[0].cpus);
        return host_run(c, H, nullptr, n, lane_of(c, plan[0]), st_out);
    }
    /* ---- several devices: LPT split, one driver thread and one worker pool per device ---- */
    const int32_t nd = (int32_t)c->children.size();
    std::vector<int64_t> weight((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        const bool good = (!H->n_samples || H->n_samples[i] > 0) && H->read_len[i] > 0 && H->n_events[i] > 0 &&
                          H->n_events[i] < (uint64_t)INT32_MAX;
        const plan_read r = make_plan(i, good ? H->read_len[i] : 0, good ? (int32_t)H->n_events[i] : 0, c->k);
        weight[(size_t)i] = r.run ? r.n_bands : 0;
    }
    std::vector<int32_t> bin_of((size_t)n);
    int rc = abea_lpt_split(weight.data(), n, nd, bin_of.data());
    if (rc) return rc;
    std::vector<std::vector<int32_t>> share((size_t)nd);
    for (int32_t i = 0; i < n; ++i) share[(size_t)bin_of[(size_t)i]].push_back(i);
    std::vector<int> rcs((size_t)nd, ABEA_OK);
    std::vector<abea_stats> sts((size_t)nd);
    std::vector<std::string> errs((size_t)nd);
    std::vector<std::thread> th;
    for (int32_t d = 0; d < nd; ++d)
        th.emplace_back([&, d]() {
            bind_this_thread(plan[(size_t)d].cpus);                       /* this thread lives for one batch */
            memset(&sts[(size_t)d], 0, sizeof(abea_stats));
            abea_ctx* dev = c->children[(size_t)d];
            rcs[(size_t)d] = host_run(dev, H, share[(size_t)d].data(), (int32_t)share[(size_t)d].size(), lane_of(dev, plan[(size_t)d]),
                                      &sts[(size_t)d]);
            if (rcs[(size_t)d]) errs[(size_t)d] = abea_last_error();       /* the message is thread-local */
        });
    for (auto& t : th) t.join();
    abea_stats st; memset(&st, 0, sizeof st);
    for (int32_t d = 0; d < nd; ++d) {
        if (rcs[(size_t)d]) return abea_fail(rcs[(size_t)d], "device %d: %s", c->children[(size_t)d]->device, errs[(size_t)d].c_str());
        if (lane_no < 0) c->children[(size_t)d]->stats = sts[(size_t)d];
        stats_add(st, sts[(size_t)d]);
    }
    st.n_devices = nd;
    st.total_ms = abea_now_ms() - t_start;
    *st_out = st;
    return ABEA_OK;
}

/* the worker pools of a context's devices at the widths of its thread plan (a multi-device parent gives every child
 * (usable CPUs - 2) / n threads): for entries that use abea_parallel_for on the children directly (abea_chain.cpp) */
int abea_pool_threads(abea_ctx* c) { return c->async && c->async->full.pool ? c->async->full.pool->threads() : 0; }

void abea_host_prepare_pools(abea_ctx* c) {
    const std::vector<host_thread_plan> plan = context_thread_plan(c);
    const host_opts opt = read_opts();
    if (c->children.empty()) {
        std::lock_guard<std::mutex> lk(c->slots_mu);
        ensure_lanes(c, plan[0], async_of(c)->n_lanes, opt.n_slots, false);
        return;
    }
    for (size_t d = 0; d < c->children.size(); ++d) {
        std::lock_guard<std::mutex> lk(c->children[d]->slots_mu);
        ensure_lanes(c->children[d], plan[d], async_of(c)->n_lanes, opt.n_slots, false);
    }
}

extern "C" int abea_align_batch_host(abea_ctx* c, const abea_host_batch* H) {
    if (!c) return abea_fail(ABEA_EINVAL, "null argument");
    int rc = check_host_batch(H);
    if (rc) return rc;
    std::lock_guard<std::mutex> api(c->api_mu);
    return abea_host_batch_locked(c, H);
}

int abea_host_batch_locked(abea_ctx* c, const abea_host_batch* H) {
    int rc = check_host_batch(H);
    if (rc) return rc;
    if (async_of(c)->n_active) return abea_fail(ABEA_EBUSY, "abea_align_batch_host: %d submitted batch(es) still in flight", c->async->n_active);
    if (H->n_reads == 0) { memset(&c->stats, 0, sizeof c->stats); return ABEA_OK; }
    abea_stats st;
    rc = run_host_batch(c, H, -1, async_of(c)->n_lanes, &st);
    if (rc) return rc;
    c->stats = st;
    return ABEA_OK;
}

/* ------------------------------------------------------------------ several host batches in flight */
extern "C" int abea_set_inflight(abea_ctx* c, int32_t n_lanes) {
    if (!c || n_lanes < 1 || n_lanes > ABEA_MAX_INFLIGHT) return abea_fail(ABEA_EINVAL, "abea_set_inflight: 1..%d lanes", ABEA_MAX_INFLIGHT);
    std::lock_guard<std::mutex> api(c->api_mu);
    abea_host_async* a = async_of(c);
    if (a->n_active) return abea_fail(ABEA_EBUSY, "abea_set_inflight: %d batch(es) in flight", a->n_active);
    a->n_lanes = n_lanes;
    return ABEA_OK;
}

extern "C" int abea_align_batch_host_submit(abea_ctx* c, const abea_host_batch* H, int32_t* ticket) {
    if (!c || !ticket) return abea_fail(ABEA_EINVAL, "null argument");
    int rc = check_host_batch(H);
    if (rc) return rc;
    std::lock_guard<std::mutex> api(c->api_mu);
    abea_host_async* a = async_of(c);
    int lane = -1;
    for (int l = 0; l < a->n_lanes; ++l) if (!a->jobs[l].active) { lane = l; break; }
    if (lane < 0) return abea_fail(ABEA_EBUSY, "abea_align_batch_host_submit: all %d lanes are busy (wait for a ticket first)", a->n_lanes);
    abea_async_job& j = a->jobs[lane];
    if (j.th.joinable()) j.th.join();
    j.H = *H; j.rc = ABEA_OK; j.err.clear(); j.active = true; j.waiting = false; ++j.gen;
    memset(&j.st, 0, sizeof j.st);
    ++a->n_active;
    const int n_lanes = a->n_lanes;
    *ticket = (int32_t)((j.gen & 0xFFFFFFu) << 3) | lane;
    abea_async_job* jp = &j;
    j.th = std::thread([c, jp, lane, n_lanes]() {
        if (jp->H.n_reads > 0) jp->rc = run_host_batch(c, &jp->H, lane, n_lanes, &jp->st);
        if (jp->rc) jp->err = abea_last_error();
    });
    return ABEA_OK;
}

extern "C" int abea_align_batch_host_wait(abea_ctx* c, int32_t ticket) {
    if (!c || ticket < 0) return abea_fail(ABEA_EINVAL, "abea_align_batch_host_wait: bad argument");
    const int lane = ticket & 7;
    abea_async_job* j = nullptr;
    {
        std::lock_guard<std::mutex> api(c->api_mu);
        abea_host_async* a = async_of(c);
        if (lane >= ABEA_MAX_INFLIGHT || !a->jobs[lane].active || (int32_t)((a->jobs[lane].gen & 0xFFFFFFu) << 3 | lane) != ticket)
            return abea_fail(ABEA_EINVAL, "abea_align_batch_host_wait: ticket %d is not in flight", ticket);
        if (a->jobs[lane].waiting)               /* a ticket is redeemed once: two joins of one thread would be undefined */
            return abea_fail(ABEA_EBUSY, "abea_align_batch_host_wait: another thread is already waiting for ticket %d", ticket);
        j = &a->jobs[lane];
        j->waiting = true;
    }
    if (j->th.joinable()) j->th.join();                      /* outside the lock: other tickets can be submitted / waited meanwhile */
    std::lock_guard<std::mutex> api(c->api_mu);
    j->active = false; j->waiting = false;
    --c->async->n_active;
    if (j->rc) return abea_fail(j->rc, "%s", j->err.c_str());
    c->stats = j->st;
    return ABEA_OK;
}

/* per-device statistics of the last host batch of a multi-device context (device = index into device_ids) */
extern "C" int abea_get_device_stats(abea_ctx* c, int32_t device, abea_stats* out) {
    if (!c || !out) return abea_fail(ABEA_EINVAL, "null argument");
    if (c->children.empty()) { if (device != 0) return abea_fail(ABEA_EINVAL, "device %d of 1", device); *out = c->stats; return ABEA_OK; }
    if (device < 0 || device >= (int32_t)c->children.size()) return abea_fail(ABEA_EINVAL, "device %d of %zu", device, c->children.size());
    *out = c->children[(size_t)device]->stats;
    return ABEA_OK;
}
