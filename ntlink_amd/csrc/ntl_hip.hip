/*
 * Host side of libntlink_hip.so: the C ABI of include/ntlink_amd.h over the HIP kernels.
 * One context = one device + one stream; all launches are asynchronous on that stream and the
 * host only synchronises where a size has to come back (totals of the offset scans).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <map>
#include <unordered_map>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ntlink_amd.h"
#include "dev_common.h"
#include "scan_kernels.h"
#include "sketch_kernels.h"
#include "sketch2_kernels.h"
#include "sketch_small_kernels.h"
#include "map_kernels.h"
#include "pack_kernels.h"
#include "synth_kernels.h"
#include "overlap_kernels.h"
#include "format_kernels.h"

#define NTL_END_PAD 4096u /* bases of padding behind the last sequence (rolling over-reads) */
#define SK_NT 256
#define NTL_MAX_W (16 * (SK_NT - 4) + 31) /* 16 k-mers per lane, a + 2 <= SK_NT - 2 halo lanes: w <= 4063 */

/* ------------------------------------------------------------------ context ------------- */

/*
 * Streams.  A context owns the MAIN stream (uploads, the emit / lookup / map / compaction kernels, downloads) and -- unless
 * NTL_PIPELINE=0 -- a second one for the WINDOW stage of a sketch (strip tables + the VALU-bound window kernels).  The window
 * stage of read batch i+1 depends on nothing the rest of batch i produces, so it is queued on its own stream and runs beside
 * batch i's latency-bound kernels; hipEvents carry the two real dependencies (window stage -> emit of the same batch; emit ->
 * the next user of the bitmask it read and cleared).  No call of the hot path waits on the host: sizes stay on the device,
 * result handles are completed lazily (sketch_finalize / mapres_finalize) when a count or a record is asked for.
 */
/* SID_P (round 4, an experiment behind NTL_PREP_STREAM=1): a third stream that only runs a sketch's PREPARATION -- the per-sequence
 * tables, the scan of the strip counts, the strip table: small dependent kernels that need nothing but the batch and sit on the window
 * stream's critical path (C3: 2.5 of 71 ms per step).  On a stream of their own they run while the previous window kernel still does
 * -- and the step gets LONGER (ntl_ctx_create); by default SID_P is the window stream. */
enum { SID_MAIN = 0, SID_W = 1, SID_P = 2, NTL_NSID = 3 };

struct ProfEntry {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    double done_ms = 0;
    uint64_t launches = 0;
};

/* a cached device block that was used on both streams: whoever takes it waits for the events of the streams it is not on */
struct XBlock {
    void *p = nullptr;
    hipEvent_t ev[NTL_NSID] = {nullptr, nullptr, nullptr}; /* nullptr: not used on that stream */
};

/* a zero-filled minimizer bitmask whose last reader (emit_kernel) cleared what the window stage had set */
struct CleanMask {
    void *p = nullptr;
    size_t bytes = 0;
    hipEvent_t clean = nullptr; /* on MAIN: the clearing kernel has run */
};

struct PinSlot { uint64_t w[8]; }; /* 64 page-locked bytes a device-side size lands in */

/* what is left of a handle that was destroyed before the device had finished its work: the event, the slot, and what to
   check in the slot once the event has passed (results nobody looked at still must not have failed silently) */
struct Zombie {
    hipEvent_t done = nullptr;
    PinSlot *slot = nullptr;
    int kind = 0;      /* 1 sketch (w[0] low = minimizer total vs cap), 2 map result (w[1] low = invariant flag) */
    uint64_t cap = 0;
    std::shared_ptr<std::atomic<float>> hitf; /* 2: where the batch's hit fraction goes (MapSums: nfound / nmx) */
    const struct ntl_index *ix = nullptr; /* the index the queued kernels read: its reference is dropped when they have run */
};

struct ntl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  /* MAIN */
    hipStream_t wstream = nullptr; /* window stage; == stream when the pipeline is off */
    hipStream_t wstream_own = nullptr; /* the second stream itself (ntl_ctx_set_pipeline switches wstream between it and MAIN) */
    hipStream_t pstream = nullptr, pstream_own = nullptr; /* a sketch's preparation (SID_P); == stream when the pipeline is off */
    bool pipelined = false;
    std::string err;
    std::string async_err;         /* first failure of work whose handle was already gone: reported by ntl_ctx_sync */
    std::string devname;
    int skw_budget = 0;            /* two streams: chunks of strips a window wavefront takes before it ends (0: resident wavefronts; tuning) */
    int n_cu = 1;                  /* compute units of the device: the grid of a kernel whose wavefronts stay resident */
    std::map<const void *, int> occ; /* kernel -> workgroups one CU holds (hipOccupancyMaxActiveBlocksPerMultiprocessor, asked once) */
    bool prof = false;
    std::map<std::string, ProfEntry> profs;
    std::vector<hipEvent_t> ev_free;    /* timing events (profiling spans) */
    std::vector<hipEvent_t> sev_free;   /* ordering events (no timing) */
    void *g4 = nullptr;                 /* device copy of the four-base init table */
    void *g8 = nullptr;                 /* device copy of the eight-base init table (1 MB) */
    std::map<int, void *> g8k;          /* k -> the two k-dependent ring forms of g8 the fast window pass reads (1 MB per k, sketch2_kernels.h) */
    std::multimap<size_t, void *> pool[NTL_NSID]; /* cached device blocks by size, per stream they were last used on */
    std::multimap<size_t, XBlock> xpool;   /* ... and those that were used on both */
    size_t pool_bytes = 0;
    size_t pool_cap = (size_t)32 << 30; /* upper bound of pool_bytes */
    /* Blocks up to NTL_SLAB_MAX_REQ are cut from slabs of NTL_SLAB_BYTES (a bump pointer; the cache above recycles them): a
       hipMalloc costs 4-9 ms of host time whatever its size, and a context's first read batch asked for seventy of them -- 0.6 s
       per context of a process's first pass (profiles/r04_first_pass.txt).  A slab block that the cache drops goes onto the slabs'
       free list (by size) and is handed out again; a slab without a live block goes back to the driver when memory runs out
       (dev_alloc), so the bound of the cache bounds the slabs too (round 5). */
    struct Slab { char *base; size_t size; size_t live; };
    std::vector<Slab> slabs;
    std::multimap<size_t, void *> slab_free;   /* dropped slab blocks by size */
    std::unordered_map<void *, size_t> slab_size; /* every slab block's TRUE size (a reused block may be up to 25 % larger than what was asked for; round 6) */
    /* slab blocks that were given back while work on them may still be queued: they join slab_free when the events recorded behind that
       work have passed (polled by dev_alloc) -- no stream is waited for (round 6, ADVICE r5: dev_free used to synchronise all three) */
    struct Limbo { void *p; size_t bytes; hipEvent_t ev[3]; };
    std::deque<Limbo> slab_limbo;
    char *slab_cur = nullptr, *slab_end = nullptr;
    std::mutex slab_mu; /* (dev_free may be asked about another context's block: index_unref) */
    std::deque<CleanMask> masks;
    void *host_tmp = nullptr;           /* page-locked bounce buffer for record downloads (grows, never shrinks) */
    size_t host_tmp_cap = 0;
    PinSlot *slots = nullptr;
    std::vector<uint32_t> slot_free;
    std::deque<Zombie> zombies;
    hipEvent_t throttle[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint64_t n_enqueued = 0;            /* sketches queued so far: the host runs at most 8 of them ahead of the device */
    hipStream_t s(int sid) const { return sid == SID_W ? wstream : (sid == SID_P ? pstream : stream); }
    int sid(int want) const
    {
        if (!pipelined) return SID_MAIN;
        if (want == SID_P && pstream == wstream) return SID_W; /* no stream of its own: the preparation is window-stage work */
        return want;
    }
};
#define NTL_NSLOTS 512u

/* Page-locked host memory.  Large blocks are an anonymous mapping on 2-MB boundaries that asks for transparent huge pages and is then
 * registered with the runtime: 6.5 ms per 146-MB staging buffer against 33-38 ms for hipHostMalloc, same DMA rate (54 GB/s) --
 * tools/pin_bench.py, profiles/r04_pin_bench.txt; a process's first pass page-locks 1.4 GB of them.  Small blocks, and whatever the
 * mapping or the registration refuses: hipHostMalloc. */
#ifndef NTL_SIM
#include <sys/mman.h>
static std::mutex g_pin_mu;
static std::map<void *, std::pair<void *, size_t>> g_pin_maps; /* registered pointer -> (mapping, its length) */
#endif

static hipError_t pin_alloc(void **out, size_t bytes)
{
#ifndef NTL_SIM
    static const bool off = getenv("NTL_PIN_HOSTMALLOC") != nullptr; /* A/B */
    if (bytes >= ((size_t)4 << 20) && !off) {
        const size_t huge = (size_t)2 << 20, len = ((bytes + huge - 1) & ~(huge - 1)) + huge;
        void *base = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base != MAP_FAILED) {
            void *p = (void *)(((uintptr_t)base + huge - 1) & ~(uintptr_t)(huge - 1));
            (void)madvise(p, len - huge, MADV_HUGEPAGE);
            if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> g(g_pin_mu);
                g_pin_maps[p] = {base, len};
                *out = p;
                return hipSuccess;
            }
            (void)hipGetLastError();
            munmap(base, len);
        }
    }
#endif
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
}

static void pin_free(void *p)
{
    if (!p) return;
#ifndef NTL_SIM
    {
        std::unique_lock<std::mutex> g(g_pin_mu);
        auto it = g_pin_maps.find(p);
        if (it != g_pin_maps.end()) {
            const std::pair<void *, size_t> m = it->second;
            g_pin_maps.erase(it);
            g.unlock();
            (void)hipHostUnregister(p);
            munmap(m.first, m.second);
            return;
        }
    }
#endif
    (void)hipHostFree(p);
}

static int host_tmp(ntl_ctx *c, size_t bytes, void **out)
{
    if (c->host_tmp_cap < bytes) {
        if (c->host_tmp) pin_free(c->host_tmp);
        c->host_tmp = nullptr; c->host_tmp_cap = 0;
        const size_t cap = bytes + bytes / 4 + 4096;
        if (pin_alloc(&c->host_tmp, cap) != hipSuccess) {
            c->host_tmp = nullptr;
            c->err = "hipHostMalloc failed";
            return NTL_ENOMEM;
        }
        c->host_tmp_cap = cap;
    }
    *out = c->host_tmp;
    return NTL_OK;
}

static int fail(ntl_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, NTL_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

/* ordering events come from a free list (creating one costs tens of microseconds) */
static hipEvent_t sev_get(ntl_ctx *c)
{
    hipEvent_t e = nullptr;
    if (!c->sev_free.empty()) { e = c->sev_free.back(); c->sev_free.pop_back(); return e; }
    /* hipEventBlockingSync: a host thread that waits on such an event sleeps until the interrupt instead of spinning -- the pair
       driver's worker threads wait for uploads and results most of the time, and on a host that grants the process 16 cores
       (the GPU boxes: 256 visible, cpu.max = 16) every spinning thread is a parser thread less */
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) return nullptr;
    return e;
}
static void sev_put(ntl_ctx *c, hipEvent_t e) { if (e) c->sev_free.push_back(e); }

/* Wait for an event the host needs NOW (a lazily completed handle): polls for a while before it blocks -- a blocking wait
 * is woken by an interrupt some tens of microseconds after the event has passed. */
static hipError_t wait_hot(hipEvent_t e)
{
    static const int spin_us = [] { const char *v = getenv("NTL_SYNC_SPIN_US"); return v ? atoi(v) : 100; }();
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipEventQuery(e);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
        }
    }
    return hipEventSynchronize(e);
}

/* waits until everything queued on MAIN so far has finished -- on an event, so that the thread sleeps (hipStreamSynchronize
   spins for a while first) */
static hipError_t main_wait(ntl_ctx *c)
{
    hipEvent_t e = sev_get(c);
    if (!e) return hipStreamSynchronize(c->stream);
    hipError_t rc = hipEventRecord(e, c->stream);
    if (rc == hipSuccess) rc = hipEventSynchronize(e);
    sev_put(c, e);
    return rc;
}

static PinSlot *slot_get(ntl_ctx *c);
static void index_unref(const struct ntl_index *ix, ntl_ctx *by);
static void slot_put(ntl_ctx *c, PinSlot *p) { if (p) c->slot_free.push_back((uint32_t)(p - c->slots)); }

/* zombies whose work has finished: check what they carried, recycle event and slot.  block: wait for the oldest one. */
static void reap(ntl_ctx *c, bool block)
{
    while (!c->zombies.empty()) {
        Zombie &z = c->zombies.front();
        hipError_t q = hipEventQuery(z.done);
        if (q == hipErrorNotReady) {
            if (!block) return;
            q = hipEventSynchronize(z.done);
            block = false;
        }
        if (q != hipSuccess && c->async_err.empty()) c->async_err = std::string("device work failed: ") + hipGetErrorString(q);
        if (q == hipSuccess && z.slot && c->async_err.empty()) {
            if (z.kind == 1 && ((uint32_t)z.slot->w[0] > z.cap || (uint32_t)(z.slot->w[1] >> 32)))
                c->async_err = "a sketch held more minimizers than its record array and was destroyed before anybody asked for its count";
            if (z.kind == 2 && (uint32_t)z.slot->w[1])
                c->async_err = "an accepted contig appeared twice in one read (bin/ntlink_utils.py:262-266)";
            if (z.kind == 2 && z.hitf) { /* MapSums: w[0] = nfound, high half of w[3] = nmx */
                const uint32_t nmx = (uint32_t)(z.slot->w[3] >> 32);
                if (nmx) z.hitf->store((float)((double)z.slot->w[0] / (double)nmx), std::memory_order_relaxed);
            }
        }
        sev_put(c, z.done);
        slot_put(c, z.slot);
        index_unref(z.ix, c);
        c->zombies.pop_front();
    }
}

static PinSlot *slot_get(ntl_ctx *c)
{
    if (c->slot_free.empty()) reap(c, true);
    if (c->slot_free.empty()) return nullptr;
    PinSlot *p = c->slots + c->slot_free.back();
    c->slot_free.pop_back();
    memset(p, 0, sizeof *p);
    return p;
}

#define NTL_SLAB_BYTES ((size_t)1 << 30)
#define NTL_SLAB_MAX_REQ ((size_t)192 << 20)

static hipError_t sync_both(ntl_ctx *c);

/* the slab a block lies in (slab_mu held), or NULL */
static ntl_ctx::Slab *slab_of(ntl_ctx *c, const void *p)
{
    for (auto &sl : c->slabs)
        if ((const char *)p >= sl.base && (const char *)p < sl.base + sl.size) return &sl;
    return nullptr;
}

/* Gives a block of dev_alloc (of `bytes`, as asked for there) back.  A single block: hipFree, which waits for the device.  A slab
   block: the same wait -- whatever is still queued on the block has run -- and then onto the slabs' free list. */
static void slab_limbo_poll(ntl_ctx *c, bool wait); /* (slab_mu held) */

static void dev_free(ntl_ctx *c, void *p, size_t bytes)
{
    if (!p) return;
    (void)bytes;
    std::lock_guard<std::mutex> g(c->slab_mu);
    ntl_ctx::Slab *sl = slab_of(c, p);
    if (!sl) { (void)hipFree(p); return; }
    /* what is still queued on the block: an event behind each of the context's streams (made here, not taken from the context's event
       list: this may be another context's thread, index_unref); no event to be had: the slow, safe way */
    ntl_ctx::Limbo L;
    L.p = p;
    auto ts = c->slab_size.find(p);
    L.bytes = ts != c->slab_size.end() ? ts->second : ((bytes + 255) & ~(size_t)255);
    hipStream_t st[3] = {c->stream, c->wstream != c->stream ? c->wstream : nullptr,
                         c->pstream != c->stream && c->pstream != c->wstream ? c->pstream : nullptr};
    bool ok = true;
    for (int i = 0; i < 3; i++) {
        L.ev[i] = nullptr;
        if (i && !st[i]) continue;
        if (hipEventCreateWithFlags(&L.ev[i], hipEventDisableTiming) != hipSuccess || hipEventRecord(L.ev[i], st[i]) != hipSuccess) { ok = false; break; }
    }
    if (!ok) {
        (void)hipGetLastError();
        for (int i = 0; i < 3; i++) if (L.ev[i]) { (void)hipEventDestroy(L.ev[i]); L.ev[i] = nullptr; }
        (void)sync_both(c);
    }
    if (sl->live) sl->live--;
    c->slab_limbo.push_back(L);
    slab_limbo_poll(c, false);
}

/* limbo blocks whose events have passed go onto the free list (wait: all of them, after waiting) */
static void slab_limbo_poll(ntl_ctx *c, bool wait)
{
    for (auto it = c->slab_limbo.begin(); it != c->slab_limbo.end();) {
        bool done = true;
        for (int i = 0; i < 3 && done; i++)
            if (it->ev[i]) {
                hipError_t q = wait ? hipEventSynchronize(it->ev[i]) : hipEventQuery(it->ev[i]);
                if (q == hipErrorNotReady) done = false;
            }
        if (!done) { ++it; continue; }
        (void)hipGetLastError();
        for (int i = 0; i < 3; i++) if (it->ev[i]) (void)hipEventDestroy(it->ev[i]);
        c->slab_free.insert({it->bytes, it->p});
        it = c->slab_limbo.erase(it);
    }
    (void)hipGetLastError(); /* (hipErrorNotReady is not an error) */
}

/* slabs without a live block go back to the driver (out of memory: dev_alloc); returns the bytes freed */
static size_t slabs_trim(ntl_ctx *c)
{
    std::lock_guard<std::mutex> g(c->slab_mu);
    slab_limbo_poll(c, true); /* (memory has run out: wait for what is in limbo, its slabs may go) */
    size_t freed = 0;
    for (size_t i = 0; i < c->slabs.size();) {
        ntl_ctx::Slab sl = c->slabs[i];
        if (sl.live) { i++; continue; }
        for (auto it = c->slab_free.begin(); it != c->slab_free.end();)
            it = ((char *)it->second >= sl.base && (char *)it->second < sl.base + sl.size) ? c->slab_free.erase(it) : std::next(it);
        if (c->slab_cur >= sl.base && c->slab_cur <= sl.base + sl.size) c->slab_cur = c->slab_end = nullptr;
        for (auto it = c->slab_size.begin(); it != c->slab_size.end();)
            it = ((char *)it->first >= sl.base && (char *)it->first < sl.base + sl.size) ? c->slab_size.erase(it) : std::next(it);
        (void)hipFree(sl.base);
        freed += sl.size;
        c->slabs.erase(c->slabs.begin() + (long)i);
    }
    return freed;
}

static void pool_drop_all(ntl_ctx *c)
{
    for (int i = 0; i < NTL_NSID; i++) {
        for (auto &kv : c->pool[i]) dev_free(c, kv.second, kv.first);
        c->pool[i].clear();
    }
    for (auto &kv : c->xpool) {
        dev_free(c, kv.second.p, kv.first);
        for (int o = 0; o < NTL_NSID; o++) sev_put(c, kv.second.ev[o]);
    }
    c->xpool.clear();
    c->pool_bytes = 0;
}

static const bool g_pool_trace = getenv("NTL_POOL_TRACE") != nullptr; /* diagnostics: every hipMalloc / hipFree of the block cache on stderr */

static int dev_alloc(ntl_ctx *c, size_t bytes, void **out)
{
#ifdef NTL_SIM
    static const bool use_slabs = false; /* the mock's blocks stay single allocations: the CPU sanitizers see every array's ends */
#else
    static const bool use_slabs = [] { const char *e = getenv("NTL_SLABS"); return !e || atoi(e) != 0; }();
#endif
    if (use_slabs && bytes <= NTL_SLAB_MAX_REQ) {
        const size_t need = (bytes + 255) & ~(size_t)255;
        {   /* a dropped slab block of this size (whatever was queued on it has run: slab_limbo_poll) */
            std::lock_guard<std::mutex> g(c->slab_mu);
            if (!c->slab_limbo.empty()) slab_limbo_poll(c, false);
            auto it = c->slab_free.lower_bound(need);
            if (it != c->slab_free.end() && it->first <= need + need / 4) {
                *out = it->second;
                if (ntl_ctx::Slab *sl = slab_of(c, it->second)) sl->live++;
                c->slab_free.erase(it);
                return NTL_OK;
            }
        }
        if (!c->slab_cur || (size_t)(c->slab_end - c->slab_cur) < need) {
            void *sl = nullptr;
            const auto ts = std::chrono::steady_clock::now();
            if (hipMalloc(&sl, NTL_SLAB_BYTES) == hipSuccess) {
                ProfEntry &pe = c->profs["hipMalloc"];
                pe.done_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count();
                pe.launches++;
                std::lock_guard<std::mutex> g(c->slab_mu);
                /* what is left of the slab before it, as a free block (else it would be lost while one of its blocks lives) */
                if (c->slab_cur && c->slab_end - c->slab_cur >= 256) {
                    c->slab_free.insert({(size_t)(c->slab_end - c->slab_cur) & ~(size_t)255, c->slab_cur});
                    c->slab_size[c->slab_cur] = (size_t)(c->slab_end - c->slab_cur) & ~(size_t)255;
                }
                c->slabs.push_back({(char *)sl, NTL_SLAB_BYTES, 0});
                c->slab_cur = (char *)sl; c->slab_end = (char *)sl + NTL_SLAB_BYTES;
            } else (void)hipGetLastError(); /* no room for a slab: single blocks as before */
        }
        if (c->slab_cur && (size_t)(c->slab_end - c->slab_cur) >= need) {
            std::lock_guard<std::mutex> g(c->slab_mu);
            *out = c->slab_cur;
            c->slab_size[c->slab_cur] = need;
            if (ntl_ctx::Slab *sl = slab_of(c, c->slab_cur)) sl->live++;
            c->slab_cur += need;
            return NTL_OK;
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(out, bytes);
    {   /* host time of the block cache's misses: ntl_prof_get(ctx, "hipMalloc") -- what a process's first pass pays once */
        ProfEntry &pe = c->profs["hipMalloc"];
        pe.done_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        pe.launches++;
    }
    if (g_pool_trace)
        fprintf(stderr, "ntl pool: hipMalloc %.1f MB -> %s in %.3f ms (cached %.1f MB of %.1f)\n", bytes / 1e6, e == hipSuccess ? "ok" : "FAILED",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), c->pool_bytes / 1e6, c->pool_cap / 1e6);
    if (e != hipSuccess) {
        /* drop the cache -- single blocks go back to the driver, slab blocks to their free list and the slabs that hold nothing
           else to the driver -- and retry once */
        pool_drop_all(c);
        (void)slabs_trim(c);
        e = hipMalloc(out, bytes);
        if (e != hipSuccess) return fail(c, NTL_ENOMEM, "hipMalloc failed");
    }
    return NTL_OK;
}

/* A device buffer that returns to the context's cache.  The cache is stream-ordered: a block goes back while kernels that
 * use it may still be queued, and is handed out again to work queued BEHIND them on the same stream (no wait).  A block that
 * was used on both streams (touch) goes back with an event per stream, and whoever takes it next waits for the other
 * stream's. */
struct DevBuf {
    ntl_ctx *c = nullptr;
    void *p = nullptr;
    size_t bytes = 0;
    mutable uint8_t used = 0; /* bit per stream id */
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    int alloc(ntl_ctx *ctx, size_t n, int sid = SID_MAIN)
    {
        release();
        c = ctx;
        sid = c->sid(sid);
        used = (uint8_t)(1u << sid);
        size_t want = n ? n : 256;
        want = (want + 255) & ~(size_t)255;
        /* Size classes (32 per power of two, at most 3 % over the request): consecutive read batches ask for arrays whose
           sizes differ in the fourth digit, and a cached block a hair smaller than the request is useless -- without classes
           every batch allocated its largest arrays anew while the cache filled with near-misses up to its bound and then
           evicted (hipFree: a device-wide wait) exactly the blocks the next batch wanted (C5: 2.6 s per step instead of 0.45). */
        if (want >= ((size_t)1 << 20)) {
            size_t step = (size_t)1 << 15;
            while ((step << 6) <= want) step <<= 1; /* step = 2^(floor(log2 want) - 5) */
            want = (want + step - 1) & ~(step - 1);
        }
        const size_t most = want + want / 4 + (1 << 20);
        /* look for a cached block; its true size is the map key */
        auto it = c->pool[sid].lower_bound(want);
        if (it != c->pool[sid].end() && it->first <= most) {
            p = it->second; bytes = it->first;
            c->pool_bytes -= it->first;
            c->pool[sid].erase(it);
            return NTL_OK;
        }
        /* A block that was used on several streams: one whose work on the OTHER streams has run by now is taken as it is; else a
           new block is made rather than this stream made to wait (the preparation of sub-batch i+1 would wait for the window
           kernel of sub-batch i, which is what the third stream is there to avoid; two or three blocks per size then go round);
           only when no memory is to be had does the taker wait. */
        auto first = c->xpool.lower_bound(want), pick = c->xpool.end();
        for (auto xt = first; xt != c->xpool.end() && xt->first <= most; ++xt) {
            bool ready = true;
            for (int o = 0; o < NTL_NSID; o++)
                if (o != sid && xt->second.ev[o] && hipEventQuery(xt->second.ev[o]) != hipSuccess) { ready = false; break; }
            if (ready) { pick = xt; break; }
        }
        (void)hipGetLastError(); /* (hipErrorNotReady is not an error) */
        int rc = NTL_OK;
        if (pick == c->xpool.end()) {
            rc = dev_alloc(c, want, &p);
            if (rc == NTL_OK) { bytes = want; return NTL_OK; }
            p = nullptr;
            if (first == c->xpool.end() || first->first > most) return rc;
            pick = first; /* out of memory: the oldest candidate, and a wait */
            c->err.clear();
        }
        XBlock &x = pick->second;
        for (int o = 0; o < NTL_NSID; o++) {
            if (o != sid && x.ev[o]) (void)hipStreamWaitEvent(c->s(sid), x.ev[o], 0);
            sev_put(c, x.ev[o]);
        }
        p = x.p; bytes = pick->first;
        c->pool_bytes -= pick->first;
        c->xpool.erase(pick);
        return NTL_OK;
    }
    void touch(int sid) const { if (c) used |= (uint8_t)(1u << c->sid(sid)); }
    void release()
    {
        if (p && c) {
            if (used == 1u || used == 2u || used == 4u) c->pool[used >> 1].insert({bytes, p});
            else {
                XBlock x;
                x.p = p;
                for (int o = 0; o < NTL_NSID; o++) {
                    if (!(used & (1u << o))) continue;
                    x.ev[o] = sev_get(c);
                    if (x.ev[o]) (void)hipEventRecord(x.ev[o], c->s(o));
                    else (void)hipStreamSynchronize(c->s(o)); /* no event to be had: the slow, safe way */
                }
                c->xpool.insert({bytes, x});
            }
            c->pool_bytes += bytes;
            /* the cache is bounded (half of the device memory unless NTL_POOL_MAX_BYTES says otherwise; a bound below the
               working set of a batch -- tens of GB for 4-Gbases HiFi batches -- turns every release into a hipFree): the
               largest blocks go first, they are the least likely to be asked for again at exactly their size */
            while (c->pool_bytes > c->pool_cap) {
                std::multimap<size_t, void *> *big = nullptr;
                for (int i = 0; i < NTL_NSID; i++)
                    if (!c->pool[i].empty() && (!big || std::prev(c->pool[i].end())->first > std::prev(big->end())->first)) big = &c->pool[i];
                if (big && (c->xpool.empty() || std::prev(big->end())->first >= std::prev(c->xpool.end())->first)) {
                    auto it = std::prev(big->end());
                    if (g_pool_trace) fprintf(stderr, "ntl pool: over the bound, hipFree %.1f MB\n", it->first / 1e6);
                    dev_free(c, it->second, it->first); /* waits for the device: safe whatever is still queued */
                    c->pool_bytes -= it->first;
                    big->erase(it);
                } else if (!c->xpool.empty()) {
                    auto it = std::prev(c->xpool.end());
                    dev_free(c, it->second.p, it->first);
                    for (int o = 0; o < NTL_NSID; o++) sev_put(c, it->second.ev[o]);
                    c->pool_bytes -= it->first;
                    c->xpool.erase(it);
                } else break;
            }
        }
        p = nullptr; bytes = 0; used = 0;
    }
    ~DevBuf() { release(); }
    template <typename T> T *as() const { return (T *)p; }
};

struct ProfSpan {
    ntl_ctx *c;
    const char *name;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    ProfSpan(ntl_ctx *ctx, const char *nm, int sid = SID_MAIN) : c(ctx), name(nm), st(ctx->s(sid))
    {
        if (!c->prof) return;
        auto get = [&](hipEvent_t &e) {
            if (!c->ev_free.empty()) { e = c->ev_free.back(); c->ev_free.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        };
        get(a); get(b);
        if (a) (void)hipEventRecord(a, st);
    }
    ~ProfSpan()
    {
        if (!c->prof || !a || !b) return;
        (void)hipEventRecord(b, st);
        ProfEntry &P = c->profs[name];
        P.spans.push_back({a, b});
        P.launches++;
    }
};

static void make_g4(uint64_t g4[256][2]);
static void make_g8(std::vector<uint64_t> &g8);

static void ctx_prime(ntl_ctx *c); /* behind the kernels' launch helpers */

extern "C" int ntl_ctx_create(int device, ntl_ctx **out)
{
    if (!out) return NTL_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return NTL_EDEVICE;
    ntl_ctx *c = new ntl_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return NTL_EDEVICE; }
    /* NTL_PIPELINE=0: one stream, every kernel behind the previous one (what a per-kernel profile wants);
       NTL_PIPELINE_PRIO: 1 (default) the MAIN stream's short latency-bound kernels get free CU slots first, 2 the window
       stage does, 0 no priorities */
    c->pipelined = true;
    if (const char *e = getenv("NTL_PIPELINE")) c->pipelined = atoi(e) != 0;
    int prio = 1;
    if (const char *e = getenv("NTL_PIPELINE_PRIO")) prio = atoi(e);
    int least = 0, greatest = 0;
    if (!c->pipelined || prio == 0 || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) prio = 0;
    hipError_t e1 = prio ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio == 1 ? greatest : least)
                         : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e1 != hipSuccess) { delete c; return NTL_EDEVICE; }
    c->wstream = c->stream;
    if (c->pipelined) {
        hipError_t e2 = prio ? hipStreamCreateWithPriority(&c->wstream, hipStreamNonBlocking, prio == 2 ? greatest : least)
                             : hipStreamCreateWithFlags(&c->wstream, hipStreamNonBlocking);
        if (e2 != hipSuccess) { c->wstream = c->stream; c->pipelined = false; }
        else c->wstream_own = c->wstream;
    }
    c->pstream = c->stream;
    /* NTL_PREP_STREAM=1 only: measured and NOT kept as the default (profiles/r04_prep_stream.jsonl: C3 75.5-76.0 ms per step against
       72.6-73.1, C5 278-283 against 259-266) -- with the window kernels back to back the lookup kernel of the other stream never
       gets the whole device for the 0.1 ms the preparation takes, and loses more (42 -> 57 ms per step) than the window stage gains */
    if (c->pipelined && getenv("NTL_PREP_STREAM") && atoi(getenv("NTL_PREP_STREAM")) != 0) {
        /* the preparation's few small kernels are wanted early: the highest priority */
        hipError_t e3 = prio ? hipStreamCreateWithPriority(&c->pstream_own, hipStreamNonBlocking, greatest)
                             : hipStreamCreateWithFlags(&c->pstream_own, hipStreamNonBlocking);
        if (e3 != hipSuccess) c->pstream_own = nullptr;
    }
    c->pstream = c->pipelined ? (c->pstream_own ? c->pstream_own : c->wstream) : c->stream;
    {
        uint64_t g4[256][2];
        make_g4(g4);
        if (hipMalloc(&c->g4, sizeof g4) != hipSuccess || hipMemcpy(c->g4, g4, sizeof g4, hipMemcpyHostToDevice) != hipSuccess) {
            delete c;
            return NTL_EDEVICE;
        }
        std::vector<uint64_t> g8;
        make_g8(g8);
        if (hipMalloc(&c->g8, g8.size() * 8) != hipSuccess ||
            hipMemcpy(c->g8, g8.data(), g8.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
            delete c;
            return NTL_EDEVICE;
        }
        if (hipHostMalloc((void **)&c->slots, NTL_NSLOTS * sizeof(PinSlot), hipHostMallocDefault) != hipSuccess) {
            delete c;
            return NTL_EDEVICE;
        }
        uint32_t nslots = NTL_NSLOTS; /* NTL_NSLOTS (tests): fewer, so that a handful of handles reaches the bound */
        if (const char *e = getenv("NTL_NSLOTS")) nslots = (uint32_t)std::min<long>(NTL_NSLOTS, std::max<long>(2, atol(e)));
        for (uint32_t i = 0; i < nslots; i++) c->slot_free.push_back(nslots - 1 - i);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->pool_cap = std::max<size_t>((size_t)prop.totalGlobalMem / 2, (size_t)256 << 20);
        if (const char *e = getenv("NTL_POOL_MAX_BYTES")) { const long long v = atoll(e); if (v >= 0) c->pool_cap = (size_t)v; }
        char buf[256];
        snprintf(buf, sizeof buf, "%s %s %d CUs %.0f GiB", prop.name, prop.gcnArchName, prop.multiProcessorCount,
                 (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
        c->devname = buf;
        c->n_cu = prop.multiProcessorCount;
    }
    ctx_prime(c);
    *out = c;
    return NTL_OK;
}

static hipError_t sync_both(ntl_ctx *c)
{
    hipError_t e = hipStreamSynchronize(c->stream);
    if (c->wstream != c->stream) { const hipError_t e2 = hipStreamSynchronize(c->wstream); if (e == hipSuccess) e = e2; }
    if (c->pstream != c->stream && c->pstream != c->wstream) { const hipError_t e3 = hipStreamSynchronize(c->pstream); if (e == hipSuccess) e = e3; }
    return e;
}

extern "C" void ntl_ctx_destroy(ntl_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_both(c);
    reap(c, true);
    pool_drop_all(c);
    for (auto &m : c->masks) { dev_free(c, m.p, m.bytes); sev_put(c, m.clean); }
    (void)hipFree(c->g4);
    (void)hipFree(c->g8);
    for (auto &kv : c->g8k) (void)hipFree(kv.second);
    {   /* (what dev_free left in limbo: its events go) */
        std::lock_guard<std::mutex> g(c->slab_mu);
        slab_limbo_poll(c, true);
    }
    for (auto &sl : c->slabs) (void)hipFree(sl.base);
    c->slabs.clear();
    if (c->host_tmp) pin_free(c->host_tmp);
    if (c->slots) (void)hipHostFree(c->slots);
    for (auto &kv : c->profs)
        for (auto &sp : kv.second.spans) { (void)hipEventDestroy(sp.first); (void)hipEventDestroy(sp.second); }
    for (auto e : c->ev_free) (void)hipEventDestroy(e);
    for (auto &e : c->throttle) sev_put(c, e);
    for (auto e : c->sev_free) (void)hipEventDestroy(e);
    if (c->wstream_own) (void)hipStreamDestroy(c->wstream_own);
    if (c->pstream_own) (void)hipStreamDestroy(c->pstream_own);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char *ntl_last_error(const ntl_ctx *c) { return c ? c->err.c_str() : "no context"; }
extern "C" const char *ntl_ctx_device_name(const ntl_ctx *c) { return c ? c->devname.c_str() : ""; }
extern "C" int ntl_ctx_pipelined(const ntl_ctx *c) { return c && c->pipelined ? 1 : 0; }

/* Switches the window stage between its own stream and MAIN at a quiet point (both streams are drained first).  bench.py
 * times the pipelined steps, then a few serial ones whose per-kernel durations are those of kernels running alone. */
extern "C" int ntl_ctx_set_pipeline(ntl_ctx *c, int on)
{
    if (!c) return NTL_EINVAL;
    (void)hipSetDevice(c->device);
    HIPCHK(c, sync_both(c));
    reap(c, true);
    if (on && !c->wstream_own) return fail(c, NTL_EINVAL, "this context was created without a window stream (NTL_PIPELINE=0)");
    c->pipelined = on != 0;
    c->wstream = on ? c->wstream_own : c->stream;
    c->pstream = on ? (c->pstream_own ? c->pstream_own : c->wstream) : c->stream;
    /* Everything queued has run (both streams were drained above), so every cached block is free on either stream: they all
       move to MAIN's cache.  Switching off, sid() maps every request to MAIN and the window-stage blocks would otherwise lie
       stranded in pool[SID_W] (counted, never handed out: the next step allocated its temporaries anew); switching on, the
       window stage's first requests miss its own cache once and are served from then on. */
    for (int i = 1; i < NTL_NSID; i++) {
        for (auto &kv : c->pool[i]) c->pool[SID_MAIN].insert(kv);
        c->pool[i].clear();
    }
    for (auto &kv : c->xpool) {
        c->pool[SID_MAIN].insert({kv.first, kv.second.p});
        for (int o = 0; o < NTL_NSID; o++) sev_put(c, kv.second.ev[o]);
    }
    c->xpool.clear();
    return NTL_OK;
}

extern "C" int ntl_ctx_sync(ntl_ctx *c)
{
    if (!c) return NTL_EINVAL;
    (void)hipSetDevice(c->device);
    HIPCHK(c, sync_both(c));
    reap(c, true);
    if (!c->async_err.empty()) {
        const std::string m = c->async_err;
        c->async_err.clear();
        return fail(c, NTL_EINTERNAL, m);
    }
    return NTL_OK;
}

extern "C" int ntl_prof_enable(ntl_ctx *c, int on)
{
    if (!c) return NTL_EINVAL;
    c->prof = on != 0;
    return NTL_OK;
}

static void prof_collect(ntl_ctx *c)
{
    (void)sync_both(c);
    for (auto &kv : c->profs) {
        for (auto &sp : kv.second.spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, sp.first, sp.second) == hipSuccess) kv.second.done_ms += ms;
            c->ev_free.push_back(sp.first);
            c->ev_free.push_back(sp.second);
        }
        kv.second.spans.clear();
    }
}

extern "C" int ntl_prof_reset(ntl_ctx *c)
{
    if (!c) return NTL_EINVAL;
    prof_collect(c);
    for (auto &kv : c->profs) { kv.second.done_ms = 0; kv.second.launches = 0; }
    return NTL_OK;
}

extern "C" int ntl_prof_get(ntl_ctx *c, const char *name, double *total_ms, uint64_t *launches)
{
    if (!c || !name) return NTL_EINVAL;
    prof_collect(c);
    auto it = c->profs.find(name);
    if (total_ms) *total_ms = it == c->profs.end() ? 0.0 : it->second.done_ms;
    if (launches) *launches = it == c->profs.end() ? 0 : it->second.launches;
    return NTL_OK;
}

/* ------------------------------------------------------------------ scan helper ---------- */

/* `batch` independent exclusive scans of equal length in one set of launches: array y is in + y*(n+1) ->
 * out + y*(n+1), out[n] of each = its sum (arrays hold n+1 entries; in may equal out).  With total_host the
 * sums also come back to the host (one stream sync); without it nothing waits. */
static int device_scan(ntl_ctx *c, const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *total_host, unsigned batch = 1,
                       uint32_t *sums_dev = nullptr, int sid = SID_MAIN)
{
    uint64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (tiles == 0) tiles = 1;
    DevBuf tile;
    int rc;
    hipStream_t st = c->s(sid);
    if ((rc = tile.alloc(c, tiles * 4 * batch, sid))) return rc;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)tiles, batch), dim3(SCAN_NT), 0, st, in, n, tile.as<uint32_t>(), n + 1, tiles);
    if (tiles <= 4096) { /* two launches: every workgroup of the second sums the tiles in front of its own */
        hipLaunchKernelGGL(scan_down_self_kernel, dim3((unsigned)tiles, batch), dim3(SCAN_NT), 0, st, in, out, n,
                           (const uint32_t *)tile.as<uint32_t>(), n + 1, tiles, sums_dev);
    } else {
        hipLaunchKernelGGL(scan_tiles_kernel, dim3(1, batch), dim3(SCAN_NT), 0, st, tile.as<uint32_t>(), tiles, out + n, n + 1, sums_dev);
        hipLaunchKernelGGL(scan_down_kernel, dim3((unsigned)tiles, batch), dim3(SCAN_NT), 0, st, in, out, n,
                           (const uint32_t *)tile.as<uint32_t>(), n + 1, tiles);
    }
    HIPCHK(c, hipGetLastError());
    if (total_host) {
        for (unsigned y = 0; y < batch; y++)
            HIPCHK(c, hipMemcpyAsync(total_host + y, out + y * (n + 1) + n, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    return NTL_OK;
}

/* ------------------------------------------------------------------ batch ---------------- */

struct ntl_batch {
    ntl_ctx *c;
    std::atomic<int> refs{1}; /* the caller's + one per sketch that may still have to be redone from it (sketch_finalize) */
    uint64_t nseq = 0, bases = 0, nruns = 0, total_gpos = 0, nwords_packed = 0;
    bool any_multi = false; /* some sequence has more than one ACGT run */
    DevBuf packed, seq_base, seq_run_first, run_start, run_len;
    DevBuf seq_len_dev;             /* u32[nseq]: the `--len` column of the batch's sequences (unset where run_len already is that) */
    const uint32_t *d_seq_len = nullptr;
    std::vector<uint32_t> seq_len;
    /* a batch whose creating call returned without waiting (the synthetic ones): recorded on MAIN behind its last kernel;
       the window stream waits for it once */
    hipEvent_t ready = nullptr;
    mutable bool w_waited = false, p_waited = false;
};

static void batch_unref(const ntl_batch *cb)
{
    ntl_batch *b = const_cast<ntl_batch *>(cb);
    if (b && --b->refs == 0) {
        (void)hipSetDevice(b->c->device);
        sev_put(b->c, b->ready);
        delete b;
    }
}

/* the window stage reads the batch on its own stream */
static void batch_on_wstream(ntl_ctx *c, const ntl_batch *b)
{
    if (!c->pipelined) return;
    if (b->ready && !b->w_waited) { (void)hipStreamWaitEvent(c->wstream, b->ready, 0); b->w_waited = true; }
    b->packed.touch(SID_W); b->seq_base.touch(SID_W); b->seq_run_first.touch(SID_W); b->run_start.touch(SID_W); b->run_len.touch(SID_W);
    if (c->pstream != c->wstream) { /* the preparation reads the sequence and run tables on its own stream */
        if (b->ready && !b->p_waited) { (void)hipStreamWaitEvent(c->pstream, b->ready, 0); b->p_waited = true; }
        b->seq_base.touch(SID_P); b->seq_run_first.touch(SID_P); b->run_start.touch(SID_P); b->run_len.touch(SID_P);
    }
}

/* Host arrays in, device layout out: the bases travel as they are (one byte each) and are packed and
 * scanned for ACGT runs on the device (pack_kernels.h).  Pinned memory from ntl_host_alloc makes the
 * copy a single DMA; anything else goes through the runtime's staging buffers. */
extern "C" int ntl_batch_create(ntl_ctx *c, const char *seqs, const uint64_t *off, uint64_t nseq, ntl_batch **out)
{
    if (!c || !out || (!seqs && nseq) || !off) return NTL_EINVAL;
    *out = nullptr;
    if (nseq >= ((uint64_t)1 << 31)) return fail(c, NTL_EINVAL, "too many sequences in one batch");
    for (uint64_t i = 0; i < nseq; i++) {
        if (off[i + 1] < off[i]) return fail(c, NTL_EINVAL, "offsets must be non-decreasing");
        if (off[i + 1] - off[i] >= 0xFFFFFFF0ull) return fail(c, NTL_EINVAL, "sequence longer than 2^32 bases");
    }
    const uint64_t o0 = nseq ? off[0] : 0;
    const uint64_t total = nseq ? off[nseq] - o0 : 0;
    std::unique_ptr<ntl_batch> b(new ntl_batch());
    b->c = c;
    b->nseq = nseq;
    b->bases = total;
    b->total_gpos = NTL_LEAD_PAD + total;
    const uint64_t nwords = (NTL_LEAD_PAD + total + NTL_END_PAD + 15) / 16 + 2;
    b->nwords_packed = nwords;
    std::vector<uint64_t> seq_base(nseq + 1);
    b->seq_len.resize(nseq);
    for (uint64_t i = 0; i <= nseq; i++) seq_base[i] = NTL_LEAD_PAD + ((i < nseq ? off[i] : off[nseq]) - o0);
    for (uint64_t i = 0; i < nseq; i++) b->seq_len[i] = (uint32_t)(off[i + 1] - off[i]);

    const uint64_t n32 = (NTL_LEAD_PAD + total + 31) / 32; /* threads = 32-position groups that hold data */
    int rc;
    (void)hipSetDevice(c->device);
    DevBuf raw, valid32, ss32, starts32, rank, any_multi;
    if ((rc = b->packed.alloc(c, nwords * 4)) || (rc = b->seq_base.alloc(c, (nseq + 1) * 8)) ||
        (rc = b->seq_run_first.alloc(c, (nseq + 1) * 4)) || (rc = raw.alloc(c, total + 64)) ||
        (rc = valid32.alloc(c, (n32 + 2) * 4)) || (rc = ss32.alloc(c, (n32 + 2) * 4)) ||
        (rc = starts32.alloc(c, (n32 + 2) * 4)) || (rc = rank.alloc(c, (n32 + 2) * 4)) || (rc = any_multi.alloc(c, 4)))
        return rc;
    uint32_t nruns = 0, multi = 0;
    DevBuf start_g, end_g;
    {
    ProfSpan span(c, "batch_pack");
    if (total) HIPCHK(c, hipMemcpyAsync(raw.p, seqs + o0, total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b->seq_base.p, seq_base.data(), (nseq + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = b->seq_len_dev.alloc(c, (nseq + 1) * 4))) return rc;
    if (nseq) HIPCHK(c, hipMemcpyAsync(b->seq_len_dev.p, b->seq_len.data(), nseq * 4, hipMemcpyHostToDevice, c->stream));
    b->d_seq_len = b->seq_len_dev.as<uint32_t>();
    HIPCHK(c, hipMemsetAsync(b->packed.p, 0, nwords * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(valid32.p, 0, (n32 + 2) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(ss32.p, 0, (n32 + 2) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(starts32.p, 0, (n32 + 2) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(rank.p, 0, (n32 + 2) * 4, c->stream)); /* counts; scanned in place */
    HIPCHK(c, hipMemsetAsync(any_multi.p, 0, 4, c->stream));
    const unsigned g32 = (unsigned)((n32 + PACK_NT - 1) / PACK_NT);
    const unsigned gseq = (unsigned)((nseq + 1 + PACK_NT - 1) / PACK_NT);
    if (n32) {
        hipLaunchKernelGGL(pack_kernel, dim3(g32), dim3(PACK_NT), 0, c->stream, raw.as<uint8_t>(), total, b->packed.as<uint32_t>(),
                           valid32.as<uint32_t>(), n32);
        hipLaunchKernelGGL(seq_mark_kernel, dim3(gseq), dim3(PACK_NT), 0, c->stream, b->seq_base.as<uint64_t>(), (uint32_t)nseq,
                           ss32.as<uint32_t>());
        hipLaunchKernelGGL(run_count_kernel, dim3(g32), dim3(PACK_NT), 0, c->stream, valid32.as<uint32_t>(), ss32.as<uint32_t>(), n32,
                           starts32.as<uint32_t>(), rank.as<uint32_t>());
        HIPCHK(c, hipGetLastError());
    }
    /* n32 + 1 counts (the last is the zero padding word) so that rank[n32] exists for seq_base[nseq] */
    if ((rc = device_scan(c, rank.as<uint32_t>(), rank.as<uint32_t>(), n32 + 1, &nruns))) return rc;
    if (nruns >= 0xFFFFFFF0u) return fail(c, NTL_EINVAL, "too many ACGT runs in one batch");
    b->nruns = nruns;
    if ((rc = b->run_start.alloc(c, ((uint64_t)nruns + 1) * 4)) || (rc = b->run_len.alloc(c, ((uint64_t)nruns + 1) * 4)) ||
        (rc = start_g.alloc(c, ((uint64_t)nruns + 1) * 8)) || (rc = end_g.alloc(c, ((uint64_t)nruns + 1) * 8)))
        return rc;
    if (nruns) {
        hipLaunchKernelGGL(run_fill_kernel, dim3(g32), dim3(PACK_NT), 0, c->stream, valid32.as<uint32_t>(), ss32.as<uint32_t>(),
                           starts32.as<uint32_t>(), rank.as<uint32_t>(), n32, start_g.as<uint64_t>(), end_g.as<uint64_t>());
        hipLaunchKernelGGL(run_finish_kernel, dim3((nruns + PACK_NT - 1) / PACK_NT), dim3(PACK_NT), 0, c->stream, start_g.as<uint64_t>(),
                           end_g.as<uint64_t>(), nruns, b->seq_base.as<uint64_t>(), (uint32_t)nseq, b->run_start.as<uint32_t>(),
                           b->run_len.as<uint32_t>());
    }
    hipLaunchKernelGGL(seq_runs_kernel, dim3(gseq), dim3(PACK_NT), 0, c->stream, starts32.as<uint32_t>(), rank.as<uint32_t>(),
                       b->seq_base.as<uint64_t>(), (uint32_t)nseq, b->seq_run_first.as<uint32_t>(), any_multi.as<uint32_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(&multi, any_multi.p, 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); /* the caller's arrays and seq_base are free again */
    b->any_multi = multi != 0;
    *out = b.release();
    return NTL_OK;
}

/* The same batch from bases that are already packed (ntl_fastx_copy_packed + ntl_fastx_runs: the parser threads pack
 * while they copy): a quarter of the bytes cross PCIe and the pack / run-table kernels are not needed. */
extern "C" int ntl_batch_create_packed(ntl_ctx *c, const uint32_t *packed, const uint64_t *off, uint64_t nseq,
                                       const uint32_t *seq_run_first, const uint32_t *run_start, const uint32_t *run_len,
                                       uint64_t nruns, ntl_batch **out)
{
    if (!c || !out || !packed || !off || !seq_run_first || (nruns && (!run_start || !run_len))) return NTL_EINVAL;
    *out = nullptr;
    if (nseq >= ((uint64_t)1 << 31)) return fail(c, NTL_EINVAL, "too many sequences in one batch");
    if (nruns >= 0xFFFFFFF0ull) return fail(c, NTL_EINVAL, "too many ACGT runs in one batch");
    if (off[0] != 0) return fail(c, NTL_EINVAL, "offsets[0] must be 0 for a packed batch");
    std::unique_ptr<ntl_batch> b(new ntl_batch());
    b->c = c;
    b->nseq = nseq;
    b->seq_len.resize(nseq);
    std::vector<uint64_t> seq_base(nseq + 1);
    bool multi = false;
    for (uint64_t i = 0; i < nseq; i++) {
        if (off[i + 1] < off[i]) return fail(c, NTL_EINVAL, "offsets must be non-decreasing");
        if (off[i + 1] - off[i] >= 0xFFFFFFF0ull) return fail(c, NTL_EINVAL, "sequence longer than 2^32 bases");
        if (seq_run_first[i + 1] < seq_run_first[i]) return fail(c, NTL_EINVAL, "seq_run_first must be non-decreasing");
        multi |= seq_run_first[i + 1] - seq_run_first[i] > 1u;
        b->seq_len[i] = (uint32_t)(off[i + 1] - off[i]);
        seq_base[i] = NTL_LEAD_PAD + off[i];
    }
    if (seq_run_first[0] != 0 || seq_run_first[nseq] != nruns) return fail(c, NTL_EINVAL, "seq_run_first does not match the run count");
    const uint64_t total = off[nseq];
    seq_base[nseq] = NTL_LEAD_PAD + total;
    b->bases = total;
    b->total_gpos = NTL_LEAD_PAD + total;
    b->nruns = nruns;
    b->any_multi = multi;
    b->nwords_packed = (NTL_LEAD_PAD + total + NTL_END_PAD + 15) / 16 + 2;
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = b->packed.alloc(c, b->nwords_packed * 4)) || (rc = b->seq_base.alloc(c, (nseq + 1) * 8)) ||
        (rc = b->seq_run_first.alloc(c, (nseq + 1) * 4)) || (rc = b->run_start.alloc(c, (nruns + 1) * 4)) ||
        (rc = b->run_len.alloc(c, (nruns + 1) * 4)))
        return rc;
    {
        ProfSpan span(c, "batch_pack");
        HIPCHK(c, hipMemcpyAsync(b->packed.p, packed, b->nwords_packed * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(b->seq_base.p, seq_base.data(), (nseq + 1) * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(b->seq_run_first.p, seq_run_first, (nseq + 1) * 4, hipMemcpyHostToDevice, c->stream));
        if ((rc = b->seq_len_dev.alloc(c, (nseq + 1) * 4))) return rc;
        if (nseq) HIPCHK(c, hipMemcpyAsync(b->seq_len_dev.p, b->seq_len.data(), nseq * 4, hipMemcpyHostToDevice, c->stream));
        b->d_seq_len = b->seq_len_dev.as<uint32_t>();
        if (nruns) {
            HIPCHK(c, hipMemcpyAsync(b->run_start.p, run_start, nruns * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(b->run_len.p, run_len, nruns * 4, hipMemcpyHostToDevice, c->stream));
        }
    }
    HIPCHK(c, main_wait(c)); /* the caller's arrays and seq_base are free again */
    *out = b.release();
    return NTL_OK;
}

/* The same from a packed stream in which the sequences need not be contiguous (ntl_fastx_parse_span: a one-pass reader puts
 * every parser thread's sequences at an upper bound of their place): sequence i = lengths[i] bases from base position
 * positions[i] of the stream (non-decreasing, non-overlapping), which spans span_positions positions in all; the words between
 * sequences are never interpreted. */
extern "C" int ntl_batch_create_packed_at(ntl_ctx *c, const uint32_t *packed, uint64_t span_positions, const uint64_t *positions,
                                          const uint32_t *lengths, uint64_t nseq, const uint32_t *seq_run_first, const uint32_t *run_start,
                                          const uint32_t *run_len, uint64_t nruns, ntl_batch **out)
{
    if (!c || !out || !packed || (nseq && (!positions || !lengths)) || !seq_run_first || (nruns && (!run_start || !run_len))) return NTL_EINVAL;
    *out = nullptr;
    if (nseq >= ((uint64_t)1 << 31)) return fail(c, NTL_EINVAL, "too many sequences in one batch");
    if (nruns >= 0xFFFFFFF0ull) return fail(c, NTL_EINVAL, "too many ACGT runs in one batch");
    if (span_positions >= 0xFFFFFFF0ull - NTL_END_PAD) return fail(c, NTL_EINVAL, "packed stream longer than 2^32 positions");
    std::unique_ptr<ntl_batch> b(new ntl_batch());
    b->c = c;
    b->nseq = nseq;
    b->seq_len.assign(lengths, lengths + nseq);
    std::vector<uint64_t> seq_base(nseq + 1);
    bool multi = false;
    uint64_t total = 0, prev_end = 0;
    for (uint64_t i = 0; i < nseq; i++) {
        if (positions[i] < prev_end || positions[i] + lengths[i] > span_positions) return fail(c, NTL_EINVAL, "sequence positions must be non-overlapping, in order and inside the stream");
        if (seq_run_first[i + 1] < seq_run_first[i]) return fail(c, NTL_EINVAL, "seq_run_first must be non-decreasing");
        multi |= seq_run_first[i + 1] - seq_run_first[i] > 1u;
        seq_base[i] = NTL_LEAD_PAD + positions[i];
        prev_end = positions[i] + lengths[i];
        total += lengths[i];
    }
    if (seq_run_first[0] != 0 || seq_run_first[nseq] != nruns) return fail(c, NTL_EINVAL, "seq_run_first does not match the run count");
    seq_base[nseq] = NTL_LEAD_PAD + span_positions;
    b->bases = total;
    b->total_gpos = NTL_LEAD_PAD + span_positions;
    b->nruns = nruns;
    b->any_multi = multi;
    b->nwords_packed = (NTL_LEAD_PAD + span_positions + NTL_END_PAD + 15) / 16 + 2;
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = b->packed.alloc(c, b->nwords_packed * 4)) || (rc = b->seq_base.alloc(c, (nseq + 1) * 8)) ||
        (rc = b->seq_run_first.alloc(c, (nseq + 1) * 4)) || (rc = b->run_start.alloc(c, (nruns + 1) * 4)) ||
        (rc = b->run_len.alloc(c, (nruns + 1) * 4)) || (rc = b->seq_len_dev.alloc(c, (nseq + 1) * 4)))
        return rc;
    {
        ProfSpan span(c, "batch_pack");
        HIPCHK(c, hipMemcpyAsync(b->packed.p, packed, b->nwords_packed * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(b->seq_base.p, seq_base.data(), (nseq + 1) * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(b->seq_run_first.p, seq_run_first, (nseq + 1) * 4, hipMemcpyHostToDevice, c->stream));
        if (nseq) HIPCHK(c, hipMemcpyAsync(b->seq_len_dev.p, b->seq_len.data(), nseq * 4, hipMemcpyHostToDevice, c->stream));
        b->d_seq_len = b->seq_len_dev.as<uint32_t>();
        if (nruns) {
            HIPCHK(c, hipMemcpyAsync(b->run_start.p, run_start, nruns * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(b->run_len.p, run_len, nruns * 4, hipMemcpyHostToDevice, c->stream));
        }
    }
    HIPCHK(c, main_wait(c)); /* the caller's arrays and seq_base are free again */
    *out = b.release();
    return NTL_OK;
}

extern "C" void ntl_batch_destroy(ntl_batch *b) { batch_unref(b); }

extern "C" int ntl_host_alloc(ntl_ctx *c, uint64_t bytes, void **out)
{
    if (!c || !out) return NTL_EINVAL;
    *out = nullptr;
    (void)hipSetDevice(c->device);
    HIPCHK(c, pin_alloc(out, bytes));
    return NTL_OK;
}

extern "C" void ntl_host_free(ntl_ctx *c, void *p)
{
    if (!c || !p) return;
    (void)hipSetDevice(c->device);
    pin_free(p);
}
extern "C" uint64_t ntl_batch_nseq(const ntl_batch *b) { return b ? b->nseq : 0; }
extern "C" uint64_t ntl_batch_bases(const ntl_batch *b) { return b ? b->bases : 0; }

/* ------------------------------------------------------------------ synthetic batches ----- */

/* device layout of a batch whose sequences are pure ACGT: every sequence is one run */
static int synth_layout(ntl_ctx *c, const uint32_t *len, uint64_t nseq, std::unique_ptr<ntl_batch> &b, std::vector<uint64_t> &seq_base)
{
    if (nseq >= ((uint64_t)1 << 31)) return fail(c, NTL_EINVAL, "too many sequences in one batch");
    b.reset(new ntl_batch());
    b->c = c;
    b->nseq = nseq;
    seq_base.resize(nseq + 1);
    b->seq_len.assign(len, len + nseq);
    uint64_t total = 0;
    for (uint64_t i = 0; i < nseq; i++) {
        if (len[i] == 0 || len[i] >= 0xFFFFFFF0u) return fail(c, NTL_EINVAL, "synthetic sequences must hold 1..2^32-17 bases");
        seq_base[i] = NTL_LEAD_PAD + total;
        total += len[i];
    }
    seq_base[nseq] = NTL_LEAD_PAD + total;
    b->bases = total;
    b->total_gpos = NTL_LEAD_PAD + total;
    b->nruns = nseq;
    b->nwords_packed = (NTL_LEAD_PAD + total + NTL_END_PAD + 15) / 16 + 2;
    b->any_multi = false;
    int rc;
    if ((rc = b->packed.alloc(c, b->nwords_packed * 4)) || (rc = b->seq_base.alloc(c, (nseq + 1) * 8)) ||
        (rc = b->seq_run_first.alloc(c, (nseq + 1) * 4)) || (rc = b->run_start.alloc(c, (nseq + 1) * 4)) ||
        (rc = b->run_len.alloc(c, (nseq + 1) * 4)))
        return rc;
    std::vector<uint32_t> iota(nseq + 1);
    for (uint64_t i = 0; i <= nseq; i++) iota[i] = (uint32_t)i;
    HIPCHK(c, hipMemcpyAsync(b->seq_base.p, seq_base.data(), (nseq + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b->seq_run_first.p, iota.data(), (nseq + 1) * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(b->run_start.p, 0, (nseq + 1) * 4, c->stream));
    if (nseq) HIPCHK(c, hipMemcpyAsync(b->run_len.p, len, nseq * 4, hipMemcpyHostToDevice, c->stream));
    b->d_seq_len = b->run_len.as<uint32_t>(); /* one run per sequence: the run lengths are the sequence lengths */
    HIPCHK(c, hipStreamSynchronize(c->stream)); /* iota and seq_base are stack/host temporaries */
    return NTL_OK;
}

extern "C" int ntl_synth_genome(ntl_ctx *c, uint64_t seed, const uint32_t *len, uint64_t nseq, ntl_batch **out)
{
    if (!c || !out || (!len && nseq)) return NTL_EINVAL;
    *out = nullptr;
    (void)hipSetDevice(c->device);
    std::unique_ptr<ntl_batch> b;
    std::vector<uint64_t> seq_base;
    int rc = synth_layout(c, len, nseq, b, seq_base);
    if (rc) return rc;
    const uint64_t nw = b->nwords_packed;
    hipLaunchKernelGGL(synth_genome_kernel, dim3((unsigned)((nw / 2 + 256) / 256)), dim3(256), 0, c->stream, b->packed.as<uint32_t>(), nw, seed);
    HIPCHK(c, hipGetLastError());
    if (c->pipelined && (b->ready = sev_get(c))) HIPCHK(c, hipEventRecord(b->ready, c->stream));
    *out = b.release();
    return NTL_OK;
}

extern "C" int ntl_synth_slices(ntl_ctx *c, const ntl_batch *src, uint64_t seed, uint64_t n, const uint32_t *src_seq,
                                const uint32_t *src_start, const uint32_t *out_len, const uint8_t *reverse,
                                double sub, double ins, double del, ntl_batch **out)
{
    if (!c || !src || !out || (n && (!src_seq || !src_start || !out_len))) return NTL_EINVAL;
    *out = nullptr;
    if (src->nruns != src->nseq || src->any_multi) return fail(c, NTL_EINVAL, "the source batch must be pure ACGT");
    if (sub < 0 || ins < 0 || del < 0 || sub > 0.5 || ins + del > 0.5) return fail(c, NTL_EINVAL, "error rates out of range");
    const bool errors = sub > 0 || ins > 0 || del > 0;
    for (uint64_t i = 0; i < n; i++) {
        if (src_seq[i] >= src->nseq) return fail(c, NTL_EINVAL, "source sequence index out of range");
        const uint64_t need = errors ? synth_span(out_len[i]) : (uint64_t)out_len[i];
        if ((uint64_t)src_start[i] + need > src->seq_len[src_seq[i]]) return fail(c, NTL_EINVAL, "slice does not fit its source sequence");
    }
    (void)hipSetDevice(c->device);
    std::unique_ptr<ntl_batch> b;
    std::vector<uint64_t> seq_base;
    int rc = synth_layout(c, out_len, n, b, seq_base);
    if (rc) return rc;
    DevBuf d_seq, d_start, d_rev;
    if ((rc = d_seq.alloc(c, (n + 1) * 4)) || (rc = d_start.alloc(c, (n + 1) * 4)) || (rc = d_rev.alloc(c, n + 1))) return rc;
    HIPCHK(c, hipMemsetAsync(b->packed.p, 0, b->nwords_packed * 4, c->stream));
    if (n) {
        HIPCHK(c, hipMemcpyAsync(d_seq.p, src_seq, n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_start.p, src_start, n * 4, hipMemcpyHostToDevice, c->stream));
        if (reverse) HIPCHK(c, hipMemcpyAsync(d_rev.p, reverse, n, hipMemcpyHostToDevice, c->stream));
        SliceArgs A;
        A.src_packed = src->packed.as<uint32_t>(); A.src_seq_base = src->seq_base.as<uint64_t>(); A.n_src = (uint32_t)src->nseq;
        A.dst_packed = b->packed.as<uint32_t>(); A.dst_seq_base = b->seq_base.as<uint64_t>();
        A.src_seq = d_seq.as<uint32_t>(); A.src_start = d_start.as<uint32_t>(); A.reverse = reverse ? d_rev.as<uint8_t>() : nullptr;
        A.n = n; A.seed = seed;
        const double S = 16777216.0; /* 24-bit draws */
        A.t_ins = (uint32_t)(ins * S); A.t_del = (uint32_t)((ins + del) * S); A.t_sub = (uint32_t)(sub * S);
        if (errors && A.t_del == 0 && A.t_sub == 0) A.t_sub = 1; /* keeps the error path (and its source span) selected */
        hipLaunchKernelGGL(synth_slices_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, A);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); /* the caller's arrays are free again */
    *out = b.release();
    return NTL_OK;
}

extern "C" int ntl_batch_download(const ntl_batch *b, char *seqs, uint64_t *off)
{
    if (!b || (!seqs && b->bases) || !off) return NTL_EINVAL;
    ntl_ctx *c = b->c;
    if (b->nruns != b->nseq || b->any_multi) return fail(c, NTL_EINVAL, "only pure-ACGT batches can be downloaded");
    (void)hipSetDevice(c->device);
    uint64_t t = 0;
    for (uint64_t i = 0; i < b->nseq; i++) { off[i] = t; t += b->seq_len[i]; }
    off[b->nseq] = t;
    if (t != b->bases) return fail(c, NTL_EINVAL, "only pure-ACGT batches can be downloaded");
    const uint64_t CH = (uint64_t)1 << 28; /* 256 Mbases of ASCII per piece */
    DevBuf tmp;
    int rc = tmp.alloc(c, std::min<uint64_t>(CH, b->bases) + 16);
    if (rc) return rc;
    for (uint64_t p0 = 0; p0 < b->bases; p0 += CH) {
        const uint64_t m = std::min<uint64_t>(CH, b->bases - p0);
        hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((m / 16 + 256) / 256)), dim3(256), 0, c->stream,
                           (const uint32_t *)b->packed.as<uint32_t>(), (uint64_t)NTL_LEAD_PAD + p0, m, tmp.as<uint8_t>());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(seqs + p0, tmp.p, m, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return NTL_OK;
}

/* ------------------------------------------------------------------ sketch --------------- */

/* ------------------------------------------------------------------ index (type) --------- */

static std::atomic<uint64_t> g_index_gen{1};

struct ntl_index {
    ntl_ctx *c;
    /* the caller's + one per PENDING sketch / map result that was queued against it: their kernels read the table, and an
       overflowed sketch is made again from it (sketch_finalize, mapres_finalize).  ntl_index_destroy only drops the caller's. */
    mutable std::atomic<int> refs{1};
    uint64_t gen = 0;                    /* identity of this index: a sketch made for it remembers the number, not the address */
    int bits = 0;
    uint64_t nslots = 0;
    mutable uint64_t size = 0;
    mutable bool size_known = false;
    /* hit fraction of the last COMPLETED batch mapped against this index: picks the probe form of the next one.  Shared with
       results that are destroyed before they complete (their zombies report when their work is done). */
    std::shared_ptr<std::atomic<float>> hit_fraction = std::make_shared<std::atomic<float>>(0.0f);
    uint32_t n_ctg = 0;
    DevBuf slots, special, ctg_len, cnt; /* cnt: device-side count of kept keys, fetched on demand */
    DevBuf tags;                         /* one byte per slot (map_kernels.h index_tag) */
    std::vector<uint32_t> h_ctg_len;     /* source of the asynchronous upload: must outlive it */
    hipEvent_t built = nullptr;          /* on the building context's MAIN stream: other contexts' streams wait for it */
};

/* what emit_kernel leaves in the sketch's page-locked slot */
struct SketchSums { uint32_t total_mx, redo_n, fb_n, list_fail; unsigned long long nfound; }; /* redo_n, fb_n: one 8-byte copy; list_fail: the
                                                                                                      strips' lists ran out of pool (StripLists) */

struct ntl_sketch {
    ntl_ctx *c;
    std::atomic<int> refs{1};             /* the caller's + one per map result that may have to be redone from it */
    uint64_t nseq = 0;
    mutable uint64_t count = 0;           /* valid once !pending */
    mutable uint64_t strips = 0, redo_strips = 0; /* diagnostics: strips of the window pass, strips that also took the exact pass */
    mutable uint64_t fallback_strips = 0;  /* ... strips the threshold pass handed to the block-minima pass */
    mutable uint64_t nfound = 0;
    mutable DevBuf records; /* MxRecord[cap] */
    mutable DevBuf mx_off;  /* u32[nseq+1] */
    mutable uint64_t cap = 0;             /* records the arrays can hold */
    /* ntl_sketch_run_indexed: the minimizers were looked up in index cand_gen while they were emitted */
    uint64_t cand_gen = 0;
    mutable DevBuf cand;    /* Cand[cap] */
    bool no_records = false; /* ntl_sketch_run_for_map: `records` stays empty */
    mutable bool from_lists = false; /* diagnostics: the last round of sketch_enqueue wrote lists */
    mutable bool no_lists = false; /* the window stage writes the bitmask whatever the window: the second round of a sketch whose lists ran out of pool */
    mutable DevBuf rpos;    /* u32[cap] beside cand: the minimizers' positions in their reads (their strands: bit 31 of Cand::meta) */
    mutable DevBuf rlen;    /* u32[nseq]: lengths of the sketched sequences (sketches made from a batch) */
    mutable DevBuf sums;    /* SketchSums on the device: total (read by the map kernels: a sketch that overflowed its arrays is left alone) */
    /* lazy completion */
    mutable bool pending = false;
    mutable hipEvent_t done = nullptr;
    mutable PinSlot *slot = nullptr;
    mutable uint64_t gen = 0;             /* bumped when the sketch had to be made again with larger arrays */
    mutable int failed = 0;               /* sticky error code of the completion */
    const ntl_batch *src = nullptr;       /* held (refs) while pending: a sketch that overflowed is redone from it */
    const ntl_index *src_ix = nullptr;
    int k = 0, w = 0;
};

static int sketch_finalize(const ntl_sketch *s);

/* The last reference to an index goes.  Dropped by the context that built it: its device blocks return to that context's cache.
 * Dropped by ANOTHER context (a worker context completing or reaping work that outlived the caller's ntl_index_destroy): the
 * owner's cache and event lists belong to the owner's thread, so the blocks are freed outright (hipFree waits for the device). */
static void index_unref(const ntl_index *cix, ntl_ctx *by)
{
    ntl_index *ix = const_cast<ntl_index *>(cix);
    if (!ix || --ix->refs != 0) return;
    (void)hipSetDevice(ix->c->device);
    if (by != ix->c) {
        for (DevBuf *b : {&ix->slots, &ix->special, &ix->ctg_len, &ix->cnt, &ix->tags}) {
            if (b->p) dev_free(ix->c, b->p, b->bytes);
            b->p = nullptr; b->bytes = 0;
        }
        if (ix->built) (void)hipEventDestroy(ix->built);
    } else sev_put(ix->c, ix->built);
    delete ix;
}

static void sketch_unref(const ntl_sketch *cs)
{
    ntl_sketch *s = const_cast<ntl_sketch *>(cs);
    if (!s || --s->refs != 0) return;
    ntl_ctx *c = s->c;
    (void)hipSetDevice(c->device);
    if (s->pending) { /* nobody asked: the device blocks go back now (stream-ordered), event and slot when the work is done */
        Zombie z;
        z.done = s->done; z.slot = s->slot; z.kind = 1; z.cap = s->cap; z.ix = s->src_ix;
        c->zombies.push_back(z);
    } else {
        sev_put(c, s->done);
        slot_put(c, s->slot);
        index_unref(s->src_ix, c);
    }
    if (s->src) batch_unref(s->src);
    delete s;
}

static uint64_t h_srol1(uint64_t x)
{
    uint64_t m = ((x & 0x8000000000000000ull) >> 30) | ((x & 0x100000000ull) >> 32);
    return ((x << 1) & 0xFFFFFFFDFFFFFFFFull) | m;
}

/* sketch_fast_kernel's keys (sketch2_kernels.h, "exact" (3)) rely on this property of the ntHash seeds: a base and its
   complement have 31-bit rings (bits 33..63) whose set bits add up to an even number */
constexpr int ring31_bits(uint64_t s) { int n = 0; for (uint64_t r = s >> 33; r; r >>= 1) n += (int)(r & 1); return n; }
static_assert((ring31_bits(0x3c8bfbb395c60474ull) + ring31_bits(0x295549f54be24456ull)) % 2 == 0 &&
              (ring31_bits(0x3193c18562a02b4cull) + ring31_bits(0x20323ed082572324ull)) % 2 == 0,
              "the ring sum of fwd and rev could be 2^31 - 1: the 32-bit window pass would need a wrap check");

static void make_tables(int k, uint64_t roll[16][2], uint64_t seed[4][2])
{
    const uint64_t S[4] = {0x3c8bfbb395c60474ull, 0x3193c18562a02b4cull, 0x20323ed082572324ull, 0x295549f54be24456ull};
    uint64_t sk[4], sck[4];
    for (int c = 0; c < 4; c++) {
        uint64_t a = S[c], b = S[3 - c];
        for (int i = 0; i < k; i++) { a = h_srol1(a); b = h_srol1(b); }
        sk[c] = a; sck[c] = b;
        seed[c][0] = S[c]; seed[c][1] = S[3 - c];
    }
    for (int in = 0; in < 4; in++)
        for (int o = 0; o < 4; o++) {
            roll[in << 2 | o][0] = S[in] ^ sk[o];
            roll[in << 2 | o][1] = sck[in] ^ S[3 - o];
        }
}

static uint64_t h_sror1(uint64_t x)
{
    uint64_t m = ((x & 0x200000000ull) << 30) | ((x & 1ull) << 32);
    return ((x >> 1) & 0xFFFFFFFEFFFFFFFFull) | m;
}

/* four-base init table (k-independent), see dev_common.h hash_init */
static void make_g4(uint64_t g4[256][2])
{
    const uint64_t S[4] = {0x3c8bfbb395c60474ull, 0x3193c18562a02b4cull, 0x20323ed082572324ull, 0x295549f54be24456ull};
    for (int b = 0; b < 256; b++) {
        uint64_t f = 0, u = 0;
        for (int j = 0; j < 4; j++) {
            const int c = (b >> (2 * j)) & 3;
            f = h_srol1(f) ^ S[c];
            u = h_sror1(u) ^ S[3 - c];
        }
        g4[b][0] = f; g4[b][1] = u;
    }
}

static void make_g8(std::vector<uint64_t> &g8)
{
    const uint64_t S[4] = {0x3c8bfbb395c60474ull, 0x3193c18562a02b4cull, 0x20323ed082572324ull, 0x295549f54be24456ull};
    g8.resize(65536 * 2);
    for (int b = 0; b < 65536; b++) {
        uint64_t f = 0, u = 0;
        for (int j = 0; j < 8; j++) {
            const int c = (b >> (2 * j)) & 3;
            f = h_srol1(f) ^ S[c];
            u = h_sror1(u) ^ S[3 - c];
        }
        g8[2 * b] = f; g8[2 * b + 1] = u;
    }
}

/* exact 64-bit pass: over all strips (multi-run strips when `multi`), or -- redo != NULL -- over the strips the fast pass flagged */
template <int C, int NT, int R0>
static void launch_mask_r0(ntl_ctx *c, const SketchArgs &A, unsigned strips, bool single, bool multi)
{
    if (R0 < 0 || A.G.r0 == R0) {
        const unsigned grid = A.redo_list ? 2048u : ((strips + 7u) & ~7u);
        if (single) hipLaunchKernelGGL((sketch_mask_kernel<C, NT, false, R0>), dim3(grid), dim3(NT), 0, c->wstream, A);
        if (multi) hipLaunchKernelGGL((sketch_mask_kernel<C, NT, true, R0>), dim3((strips + 7u) & ~7u), dim3(NT), 0, c->wstream, A);
        return;
    }
    if constexpr (R0 >= 0 && R0 + 1 < C) launch_mask_r0<C, NT, R0 + 1>(c, A, strips, single, multi);
}

template <int C>
static void launch_mask(ntl_ctx *c, const SketchArgs &A, unsigned strips, bool single, bool multi, int nt)
{
    /* 16 k-mers per lane: one instantiation per strip width, R0 at run time (this is the redo / multi-run pass since the 32-bit
       window pass took over the common case); 4 and 1 k-mers per lane (w < 16): R0 as a template parameter */
    if (C == 16 && nt == 128) launch_mask_r0<C, 128, -1>(c, A, strips, single, multi);
    else if (C == 16) launch_mask_r0<C, SK_NT, -1>(c, A, strips, single, multi);
    else launch_mask_r0<C, SK_NT, 0>(c, A, strips, single, multi);
}

/* workgroups of `threads` lanes of a kernel that one CU holds at once */
template <typename K>
static int occupancy_blocks(K kern, int threads)
{
#ifdef NTL_SIM
    (void)kern; (void)threads;
    return 2;
#else
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, 0) != hipSuccess || n < 1) n = 1;
    return n;
#endif
}

/* What a context would otherwise do inside its first sketch call, done when it is made: the first slab of device memory and
   the occupancy figures of the resident window kernels (the query loads the library's code object onto the device: some
   ten milliseconds, once per process).  Best effort: whatever fails here is tried again where it is needed. */
__global__ void ntl_noop_kernel() {}

static void ctx_prime(ntl_ctx *c)
{
#ifndef NTL_SIM
    hipLaunchKernelGGL(ntl_noop_kernel, dim3(1), dim3(64), 0, c->stream); /* the library's code object goes onto the device with its first launch */
    (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
    {
        DevBuf first;
        (void)first.alloc(c, 1 << 20);
    }
    c->occ[(const void *)sketch_wave_kernel<8, 11, 4>] = occupancy_blocks(sketch_wave_kernel<8, 11, 4>, 512);
    c->occ[(const void *)sketch_wave_kernel<8, 15, 6>] = occupancy_blocks(sketch_wave_kernel<8, 15, 6>, 512);
    c->occ[(const void *)sketch_wave_kernel<8, 19, 8>] = occupancy_blocks(sketch_wave_kernel<8, 19, 8>, 512);
    c->occ[(const void *)sketch_wave_kernel<4, 19, 8>] = occupancy_blocks(sketch_wave_kernel<4, 19, 8>, 256);
    c->err.clear();
#else
    (void)c;
#endif
}

/* does the window pass of this sketch run as sketch_wave_kernel (one wavefront per strip)?  Where a lane's first k-mer lies in its own
   64 bases and a strip's candidates fit its list; NTL_SKETCH_WAVE=0: the workgroup-per-strip form (A/B, tests).  Read per call: the
   tests switch it inside one process. */
static int wave_form(const Sketch2Args &B, int nt)
{
    const char *we = getenv("NTL_SKETCH_WAVE");
    const int wave = we ? atoi(we) : 1;
    const double per_strip = 4096.0 * (double)B.thresh / 4294967296.0; /* candidates a strip is expected to hold */
    /* (a + 2 <= 16, w <= 255, is the workgroup-per-strip passes' limit, not this kernel's: a window only enters its scans as a distance.
       Round 6: up to the block-minima pass's w <= 1135, which decides the strips it gives up) */
    if (nt == 256 && B.thresh && B.A.G.a + 2 <= SK2_PAD && (B.dbg & ~24) == 0 && wave && B.A.G.k <= 64 && per_strip <= 440.0) return wave;
    return 0;
}

/* fast 32-bit pass over the single-run strips (sketch2_kernels.h) */
template <int NT, int R0>
static void launch_fast_r0(ntl_ctx *c, const Sketch2Args &B, unsigned strips)
{
    if (B.A.G.r0 == R0) {
        const dim3 grid((strips + 7u) & ~7u);
        /* the 20-KB variants hold 128 searched windows per strip: about NWO / (w + 1) are expected (38 at w = 100) */
        /* NTL_SKETCH_LANES=1: sketch_lanes_kernel, the variant that walks only the lanes whose minimum can change -- 30 % fewer
           VALU instructions, but its single-wavefront phases halve the number of runnable wavefronts per SIMD and the launch
           takes 5-6 % LONGER (profiles/r03*_lanes*): an experiment that is kept for the record, not the default */
        const char *le = getenv("NTL_SKETCH_LANES"); /* read per call: the tests switch it inside one process */
        const int lanes = le ? atoi(le) : 0;
        /* the window pass on threshold-sparsified windows where the geometry allows it (B.thresh != 0) */
        if (NT == 256 && B.thresh && (B.A.G.a + 2 <= 16 || wave_form(B, NT)) && (B.dbg & ~24) == 0) { /* (ablation bits 8 and 16 exist in this kernel too) */
            /* one wavefront per strip, 64 k-mers per lane (sketch_wave_kernel, round 4) where a lane's first k-mer lies in its own
               64 bases and a strip's candidates fit its list; NTL_SKETCH_WAVE=0: the workgroup-per-strip form (A/B, tests) */
            const int wave = wave_form(B, NT);
            const double per_strip = 4096.0 * (double)B.thresh / 4294967296.0; /* candidates a strip is expected to hold */
            if (wave) {
                /* resident wavefronts that walk over their strips: as many workgroups as the device holds at once (a multiple of 8:
                   one share of the strips per XCD).  <wavefronts per workgroup, staging slots per lane, scan rounds>: the slots hold a
                   lane's 64 p candidates + 4.5 sigma, the list (64 per round) a strip's 4096 p + 4 sigma; what does not fit is given up */
                auto go = [&](auto kern, unsigned threads, unsigned beside = 16u) {
                    int &per_cu = c->occ[(const void *)kern];
                    if (!per_cu) per_cu = occupancy_blocks(kern, (int)threads);
                    /* Beside the other stream's kernels (two streams: the lookup / map kernels of the previous batch) the resident wavefronts
                       take HALF the CU's 32 slots: with more, those kernels' workgroups wait for slots that never come free before the
                       launch ends, and the step is as long as on one stream (C3, profiles/r04_window_grid_sweep.json: 16 wavefronts per
                       CU 77.6 ms per step, 24 89.5, 32 89.4; alone the launch takes 2.65 ms at 16 against 2.09 at 32). */
                    int use = c->pipelined ? std::min(per_cu, (int)(beside / (threads / 64u))) : per_cu;
                    if (const char *e = getenv("NTL_SKW_WGS_PER_CU")) use = std::max(1, std::min(per_cu, atoi(e))); /* tuning */
                    unsigned wgs = (unsigned)std::max(1, use) * (unsigned)std::max(1, c->n_cu);
                    wgs = std::min(wgs, (strips + threads / 64u - 1u) / (threads / 64u));
                    wgs = (wgs + 7u) & ~7u;
                    /* Two streams, NTL_SKW_BUDGET chunks per wavefront: short-lived workgroups, as many as it takes, that fill what
                       the other stream's kernels leave free (those have the higher stream priority and a bounded number of
                       resident workgroups: sketch_enqueue, emit) -- and the whole CU while that stream has nothing to run. */
                    Sketch2Args Bq = B;
                    int budget = c->pipelined ? c->skw_budget : 0;
                    if (const char *e = getenv("NTL_SKW_BUDGET")) budget = atoi(e);
                    if (budget >= 2) {
                        const unsigned per_xcd_chunks = (((strips + 7u) >> 3) + SKW_CHUNK - 1u) / SKW_CHUNK;
                        const unsigned per_wg = (threads / 64u) * (unsigned)budget;
                        wgs = 8u * ((per_xcd_chunks + per_wg - 1u) / per_wg);
                        Bq.chunk_budget = (uint32_t)budget;
                    }
                    ProfSpan sp(c, "sketch_wave", SID_W); /* the window kernel alone (the span "sketch_mask" around this function also holds the block-minima pass) */
                    hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, c->wstream, Bq);
                };
                if (per_strip <= 175.0) { /* w >= 235 at ten candidates per window */
                    /* (lists, round 5: the other stream's emit_list_kernel keeps to two resident workgroups per CU -- sketch_enqueue --
                       and the window stage takes 24 of the 32 wavefront slots: C3 70.7 ms per step against 75.9 at 16 and 82.5 at
                       32, profiles/r05_share_sweep_C3.jsonl) */
                    const unsigned beside = B.A.Ls.cnt ? 24u : 16u;
                    if (wave == 4) go(sketch_wave_kernel<4, 11, 4>, 256u, beside);
                    else if (wave == 16) go(sketch_wave_kernel<16, 11, 4>, 1024u);
                    else go(sketch_wave_kernel<8, 11, 4>, 512u, beside);
                } else if (per_strip <= 300.0) { /* w >= 137 */
                    go(sketch_wave_kernel<8, 15, 6>, 512u);
                } else {                         /* w >= 94 */
                    /* (dense sketches: the other stream's lookup and map kernels are the longer half of a step, and twelve resident
                       wavefronts per CU beside them make the shortest step -- C5: 8 / 12 / 16 / 24: 304.6 / 288.5 / 315.1 / 318.7 ms) */
                    if (wave == 4 || (c->pipelined && wave != 8)) go(sketch_wave_kernel<4, 19, 8>, 256u, 12u);
                    else go(sketch_wave_kernel<8, 19, 8>, 512u);
                }
                if (B.A.G.a + 2 <= 16) hipLaunchKernelGGL((sketch_fast_list_kernel<256, R0>), dim3(std::min(strips, 4096u)), dim3(256), 0, c->wstream, B, (const uint32_t *)B.fb_list, (const uint32_t *)B.fb_count);
                else hipLaunchKernelGGL((sketch_fast_list_kernel<256, R0, true>), dim3(std::min(strips, 4096u)), dim3(256), 0, c->wstream, B, (const uint32_t *)B.fb_list, (const uint32_t *)B.fb_count);
                return;
            }
            const char *de = getenv("NTL_SKETCH_THRESH_DIRECT"); /* 1: the variant without staged keys for the large windows too (A/B) */
            const int direct = de ? atoi(de) : 0;
            if (direct || 4096.0 * (double)B.thresh / 4294967296.0 > 340.0) hipLaunchKernelGGL((sketch_thresh_kernel<256, true>), grid, dim3(256), 0, c->wstream, B);
            else hipLaunchKernelGGL((sketch_thresh_kernel<256, false>), grid, dim3(256), 0, c->wstream, B);
            /* what it gave up (a window without a candidate: 0.7 % of the strips): the block-minima pass over that list */
            hipLaunchKernelGGL((sketch_fast_list_kernel<256, R0>), dim3(std::min(strips, 4096u)), dim3(256), 0, c->wstream, B, (const uint32_t *)B.fb_list, (const uint32_t *)B.fb_count);
            return;
        }
        if (B.A.G.a + 2 <= 16 && B.A.G.w >= 64 && B.A.G.a >= 2 && lanes && B.dbg == 0) hipLaunchKernelGGL((sketch_lanes_kernel<NT, R0>), grid, dim3(NT), 0, c->wstream, B);
        else if (B.A.G.a + 2 <= 16 && B.A.G.w >= 64) hipLaunchKernelGGL((sketch_fast_kernel<NT, R0, false>), grid, dim3(NT), 0, c->wstream, B);
        else hipLaunchKernelGGL((sketch_fast_kernel<NT, R0, true>), grid, dim3(NT), 0, c->wstream, B);
        return;
    }
    if constexpr (R0 + 1 < 16) launch_fast_r0<NT, R0 + 1>(c, B, strips);
}

/* a zero-filled bitmask of at least `bytes` for the window stage (stream sid): one that an earlier emit kernel cleared
   behind itself, or a new one.  With the pipeline on, three masks rotate, so that the window stage of batch i+1 never
   waits for the emit kernel of batch i. */
static int mask_take(ntl_ctx *c, size_t bytes, int sid, CleanMask *out)
{
    const size_t want = (bytes + 255) & ~(size_t)255;
    const size_t keep = c->pipelined ? 2 : 0;
    if (c->masks.size() > keep) {
        for (size_t i = 0; i < c->masks.size() - keep; i++) {
            CleanMask m = c->masks[i];
            if (m.bytes >= want && m.bytes <= want + want / 4 + (1 << 20)) {
                c->masks.erase(c->masks.begin() + (long)i);
                if (m.clean && c->s(sid) != c->stream) (void)hipStreamWaitEvent(c->s(sid), m.clean, 0);
                sev_put(c, m.clean);
                m.clean = nullptr;
                *out = m;
                return NTL_OK;
            }
        }
    }
    while (c->masks.size() > 4) { /* other batch sizes came and went */
        dev_free(c, c->masks.front().p, c->masks.front().bytes);
        sev_put(c, c->masks.front().clean);
        c->masks.pop_front();
    }
    CleanMask m;
    int rc = dev_alloc(c, want, &m.p);
    if (rc) return rc;
    m.bytes = want;
    HIPCHK(c, hipMemsetAsync(m.p, 0, want, c->s(sid)));
    *out = m;
    return NTL_OK;
}

/* does the window pass of this window size run as sketch_small_kernel (2 <= w <= 15: the stages around `pair`)?  NTL_SKETCH_SMALL=0, or any
   setting of the k-mers-per-lane knob: the round-1 forms with four / one k-mer per lane (A/B, tests).  Read per call. */
static bool small_window_form(int w)
{
    if (w < 2 || w > 15 || getenv("NTL_SKETCH_C")) return false;
    const char *e = getenv("NTL_SKETCH_SMALL");
    return !e || atoi(e) != 0;
}

template <int W>
static void launch_small_w(ntl_ctx *c, const SketchArgs &A, unsigned strips, bool multi)
{
    if (A.G.w == W) {
        const dim3 grid((strips + 7u) & ~7u);
        hipLaunchKernelGGL((sketch_small_kernel<W, false>), grid, dim3(256), 0, c->wstream, A);
        if (multi) hipLaunchKernelGGL((sketch_small_kernel<W, true>), grid, dim3(256), 0, c->wstream, A);
        return;
    }
    if constexpr (W < 15) launch_small_w<W + 1>(c, A, strips, multi);
}

#ifndef NTL_EMIT_DENSE_CAP
#define NTL_EMIT_DENSE_CAP 8192 /* emit_kernel's positions per round for the dense sketches of the small windows (sketch_kernels.h) */
#endif

/* geometry of the window pass for (k, w): k-mers per lane, lanes per strip */
static int sketch_geometry(ntl_ctx *c, int k, int w, SketchGeom &G, int &C, int &nt)
{
    if (k < 1 || k > 4096 || w < 1) return fail(c, NTL_EINVAL, "k must be in 1..4096 and w >= 1");
    C = w >= 16 ? 16 : (w >= 4 ? 4 : 1);
    G.k = k; G.w = w;
    if (small_window_form(w)) { /* sketch_small_kernel: 256 lanes x 16 k-mers, the minima of the 16 windows a lane starts in registers */
        C = 16; nt = SK_NT;
        G.a = 0; G.r0 = 0;
        G.LW = nt - 1;          /* the last lane only supplies the w - 1 elements behind the last window */
        G.NWO = G.LW * C - 1;
        return NTL_OK;
    }
    if (const char *e = getenv("NTL_SKETCH_C")) { /* tuning knob: k-mers per lane (16, 4, 1) */
        const int v = atoi(e);
        if ((v == 16 || v == 4 || v == 1) && w >= v) C = v;
    }
    G.a = (w - C) / C; G.r0 = (w - C) % C;
    /* lanes per strip: 128-lane strips waste fewer lanes on the last strip of a ~10 kb read, 256-lane strips fewer on the halo
       (a+2 lanes) and fit eight workgroups of the 32-bit window pass on a CU.  Measured with that pass: 256 lanes +6 % at w=100
       (10 kb reads 675 -> 714 Gbases/s, 20 kb reads 696 -> 747); 128 lanes stay for the small windows (w < 64) */
    nt = (C == 16 && w < 64) ? 128 : SK_NT;
    if (const char *e = getenv("NTL_SKETCH_NT")) { /* tuning knob; ignored where it leaves no lane to own a window */
        const int v = atoi(e);
        if ((v == 256 || (v == 128 && C == 16)) && v - (G.a + 2) >= 2) nt = v;
    }
    G.LW = nt - (G.a + 2);
    if (G.LW < 2 && C != 16) { /* a tuning knob (NTL_SKETCH_C) made the strip too short for this window: back to the default */
        C = 16;
        G.a = (w - C) / C; G.r0 = (w - C) % C;
        nt = SK_NT;
        G.LW = nt - (G.a + 2);
    }
    G.NWO = G.LW * C - 1;
    if (G.LW < 2 || G.NWO < 1) {
        char msg[160];
        snprintf(msg, sizeof msg, "window size w=%d is beyond this build's limit of w <= %d (a strip of %d x 16 k-mers must hold two windows' lanes)",
                 w, NTL_MAX_W, SK_NT);
        return fail(c, NTL_EINVAL, msg);
    }
    return NTL_OK;
}

/* Queues one sketch: the window stage on the window stream, count + emit (+ index lookup) on MAIN behind it.  Nothing
 * waits; the minimizer total lands in the sketch's page-locked slot and s->done is recorded behind it.  `cap` = records the
 * arrays hold: a guess from the expected density (sketch_finalize makes the sketch again if the batch was denser), or the
 * exact total on that second round. */
static int sketch_enqueue(ntl_ctx *c, const ntl_batch *b, int k, int w, const ntl_index *ix, ntl_sketch *s, uint64_t cap)
{
    SketchGeom G;
    int C, nt, rc;
    if ((rc = sketch_geometry(c, k, w, G, C, nt))) return rc;
    const uint64_t nseq = b->nseq;
    const int wsid = c->sid(SID_W), psid = c->sid(SID_P);
    hipStream_t ws = c->s(wsid), ms = c->stream, ps = c->s(psid);
    /* the host runs at most 8 sketches ahead of the device */
    {
        hipEvent_t &t = c->throttle[c->n_enqueued & 7u];
        if (t) (void)hipEventSynchronize(t);
        reap(c, false);
    }
    DevBuf run_n, run_ord, seq_M, nstrips, strip_first, tile, redo, strip_tab;
    /* nstrips / strip_first hold nseq+1 entries: the scan leaves the total behind the last one */
    const uint64_t nmask = (b->total_gpos + 31) / 32 + 1;
    /* the preparation's arrays: made on its stream (psid), and those the window stage reads marked as used there too */
    if ((rc = run_n.alloc(c, (b->nruns + 1) * 4, psid)) || (rc = run_ord.alloc(c, (b->nruns + 1) * 4, psid)) ||
        (rc = seq_M.alloc(c, (nseq + 1) * 4, psid)) || (rc = nstrips.alloc(c, (nseq + 1) * 4, psid)) ||
        (rc = strip_first.alloc(c, (nseq + 2) * 4, psid)) || (rc = s->mx_off.alloc(c, (nseq + 1) * 4)) ||
        (rc = s->sums.alloc(c, sizeof(SketchSums)))) {
        return rc;
    }
    batch_on_wstream(c, b);
    SeqTables T;
    T.packed = b->packed.as<uint32_t>(); T.seq_base = b->seq_base.as<uint64_t>();
    T.seq_run_first = b->seq_run_first.as<uint32_t>(); T.run_start = b->run_start.as<uint32_t>();
    T.run_len = b->run_len.as<uint32_t>(); T.nseq = (uint32_t)nseq;
    /* upper bound of the number of strips from the host-side lengths (exact for sequences without
       non-ACGT bytes): the grid is sized without waiting for the device */
    uint64_t ub_strips = 0;
    for (uint64_t i = 0; i < nseq; i++) {
        const uint64_t len = b->seq_len[i];
        if (len + 2 > (uint64_t)k + (uint64_t)w) ub_strips += (len - k - w + 2 + (uint64_t)G.NWO - 1) / (uint64_t)G.NWO;
    }
    if (ub_strips >= 0x7FFFFFFFull) return fail(c, NTL_EINVAL, "batch too large: too many strips");
    /* Which window pass, and what it writes.  32-bit fast pass + exact pass over what it flags; the exact pass alone for small windows
       / huge k.  Where the fast pass is sketch_wave_kernel (94 <= w <= 1135, k <= 64: every window ntLink is run with) the passes write
       per-strip LISTS of minimizers (sketch_kernels.h, StripLists) and emit_list_kernel reads those; everywhere else, a bitmask of one
       bit per base and emit_kernel.  NTL_SKETCH_LISTS=0: the bitmask everywhere (A/B, tests). */
    const bool small = small_window_form(w);
    bool fast = C == 16 && k <= 16 * SK2_QMAX && G.a + 2 <= SK2_PAD && !small;
    if (const char *e = getenv("NTL_SKETCH_FAST")) fast = fast && atoi(e) != 0; /* 0: exact pass only (A/B, tests) */
    Sketch2Args B;
    memset(&B, 0, sizeof B);
    B.A.G = G;
    if (fast) {   /* sketch_thresh_kernel (threshold-sparsified windows) where a strip's candidate list fits: keys below
                     T = 2^32 * cpw / w are candidates, cpw = 10 of them per window -- fewer and more strips have a window without
                     one (they take the exact pass: 0.7 % at 10, 2 % at 9), more and the list work grows (profiles/r03p_*).
                     NTL_SKETCH_THRESH=0: sketch_fast_kernel everywhere; = x: x candidates per window.  Read per call: the tests
                     switch it inside one process. */
        const char *e = getenv("NTL_SKETCH_THRESH");
        double cpw = e ? atof(e) : 10.0;
        if (cpw == 1.0) cpw = 10.0;
        /* a strip's expected 4096 cpw / w candidates must fit the list with room for their spread: 402 entries beside
           the staged keys (w >= 121 at 10 per window), 680 without them (sketch_thresh_kernel<.., DIRECT>: w >= 71) */
        if (cpw > 0 && nt != 128 && G.a + 2 <= SK2_PAD && 4096.0 * cpw / w <= 580.0) /* (w > 255: sketch_wave_kernel only, wave_form) */
            B.thresh = (uint32_t)std::min(4294967295.0, 4294967296.0 * cpw / w);
        if (const char *e2 = getenv("NTL_SKETCH_ABLATE")) B.dbg = atoi(e2); /* tools/sketch_bench.py only: results are wrong */
        if (const char *e2 = getenv("NTL_SKETCH_FORCE_REDO")) B.force_redo = atoi(e2); /* tests: every strip takes both passes */
    }
    StripLists Ls;
    memset(&Ls, 0, sizeof Ls);
    DevBuf lcnt, lent, loff;
    bool lists = fast && nseq && ub_strips && !s->no_lists && wave_form(B, nt) != 0;
    if (const char *e = getenv("NTL_SKETCH_LISTS")) lists = lists && atoi(e) != 0;
    if (lists) {
        /* a slot holds the expected 2 NWO / (w + 1) minimizers of a strip with room for their spread (what does not fit -- low-complexity
           sequence -- lives in the pool behind the slots); the pool: 32 entries per strip, at least a million */
        const double mean = 2.0 * (double)G.NWO / (double)(w + 1);
        uint64_t slot = ((uint64_t)(mean * 1.25 + 24.0) + 15u) & ~(uint64_t)15u;
        if (const char *e = getenv("NTL_LIST_SLOT")) slot = std::max<uint64_t>(1, (uint64_t)atoll(e)); /* tests: force strips into the pool */
        uint64_t pool = std::max<uint64_t>((uint64_t)1 << 20, 32 * ub_strips);
        if (const char *e = getenv("NTL_LIST_POOL")) pool = (uint64_t)atoll(e);                      /* tests: make the pool run out */
        if ((ub_strips + 1) * slot + pool >= 0xFFFFFFF0ull) lists = false; /* (entries are addressed with 32 bits) */
        else {
            Ls.slot = (uint32_t)slot; Ls.ovf_base = (uint32_t)((ub_strips + 1) * slot); Ls.ovf_cap = (uint32_t)pool;
            /* cnt[ub_strips + 1] (+ the two control words behind it), zeroed: a strip nobody lists has none */
            if ((rc = lcnt.alloc(c, (ub_strips + 4) * 4, wsid)) || (rc = lent.alloc(c, ((ub_strips + 1) * slot + pool) * 4, wsid)) ||
                (rc = loff.alloc(c, (ub_strips + 2) * 4))) return rc;
            lcnt.touch(SID_MAIN); lent.touch(SID_MAIN);
            Ls.cnt = lcnt.as<uint32_t>(); Ls.ent = lent.as<uint32_t>(); Ls.ctl = Ls.cnt + ub_strips + 2;
            lcnt.touch(SID_P); /* (zeroed by strip_table_kernel, on the preparation's stream) */
        }
    }
    /* the fast pass's counters and lists: [0] strips for the exact pass, [1] strips the threshold pass gave up, then the eight chunk
       counters of sketch_wave_kernel (one per XCD's share of the strips, 64 bytes apart), then the two lists.  Zeroed -- with the
       strips' counts -- by strip_table_kernel: one launch instead of three fills on the window stream's critical path (round 5) */
    StripZero Z;
    memset(&Z, 0, sizeof Z);
    const uint64_t redo_head = 16 + 8 * 16; /* words in front of the lists */
    if (fast && ub_strips) {
        if ((rc = redo.alloc(c, (redo_head + 2 * ub_strips + 4) * 4, wsid))) return rc;
        redo.touch(SID_P);
        Z.head = redo.as<uint32_t>(); Z.nhead = (uint32_t)redo_head;
        Z.cnt = Ls.cnt; Z.ncnt = Ls.cnt ? (uint32_t)(ub_strips + 4) : 0u;
    }
    CleanMask mask;
    if (!lists && (rc = mask_take(c, nmask * 4, wsid, &mask))) return rc;
    struct MaskGuard { /* error paths: the mask is not known to be clean any more */
        ntl_ctx *c; CleanMask *m;
        ~MaskGuard() { if (m->p) { dev_free(c, m->p, m->bytes); m->p = nullptr; } }
    } mask_guard{c, &mask};
    if ((rc = strip_tab.alloc(c, (ub_strips + 1) * sizeof(StripInfo), psid))) return rc;
    DevBuf strip_lite;
    if ((rc = strip_lite.alloc(c, (ub_strips + 1) * sizeof(StripLite), psid))) return rc;
    run_n.touch(SID_W); run_ord.touch(SID_W); seq_M.touch(SID_W); strip_tab.touch(SID_W); strip_lite.touch(SID_W);
    SketchSums *dsums = s->sums.as<SketchSums>();
    HIPCHK(c, hipMemsetAsync(dsums, 0, sizeof(SketchSums), ms));
    {
        ProfSpan sp(c, "sketch_meta", psid);
        if (nseq) {
            KTables K;
            K.run_n = run_n.as<uint32_t>(); K.run_ord = run_ord.as<uint32_t>();
            K.seq_M = seq_M.as<uint32_t>(); K.seq_nstrips = nstrips.as<uint32_t>();
            hipLaunchKernelGGL(seq_meta_kernel, dim3((unsigned)((nseq + 255) / 256)), dim3(256), 0, ps, T, K, k, w, G.NWO);
            HIPCHK(c, hipGetLastError());
            if ((rc = device_scan(c, nstrips.as<uint32_t>(), strip_first.as<uint32_t>(), nseq, nullptr, 1, nullptr, psid))) return rc;
            hipLaunchKernelGGL(strip_table_kernel, dim3((unsigned)((ub_strips + 255) / 256 + 1)), dim3(256), 0, ps, T,
                               (const uint32_t *)run_n.as<uint32_t>(), (const uint32_t *)run_ord.as<uint32_t>(),
                               (const uint32_t *)seq_M.as<uint32_t>(), (const uint32_t *)strip_first.as<uint32_t>(), G.NWO,
                               C * nt, strip_tab.as<StripInfo>(), (uint32_t)ub_strips + 1u, strip_lite.as<StripLite>(), Z);
            HIPCHK(c, hipGetLastError());
        }
    }
    if (ps != ws && nseq) { /* preparation -> window stage */
        hipEvent_t e = sev_get(c);
        if (!e) return fail(c, NTL_EDEVICE, "hipEventCreate failed");
        HIPCHK(c, hipEventRecord(e, ps));
        HIPCHK(c, hipStreamWaitEvent(ws, e, 0));
        sev_put(c, e); /* the wait holds what it needs; the handle may be recorded again */
    }
    if (ub_strips) {
        SketchArgs A;
        A.T = T;
        A.run_n = run_n.as<uint32_t>(); A.run_ord = run_ord.as<uint32_t>(); A.seq_M = seq_M.as<uint32_t>();
        A.strip_lite = strip_lite.as<StripLite>();
        A.strip_tab = strip_tab.as<StripInfo>(); A.nstrips = (uint32_t)ub_strips; A.mask = (uint32_t *)mask.p; A.G = G;
        make_tables(k, A.roll_tab, A.seed_tab);
        A.g4 = (const uint64_t (*)[2])c->g4;
        A.g8 = (const uint64_t (*)[2])c->g8;
        A.redo_list = nullptr; A.redo_count = nullptr;
        A.Ls = Ls;
        if (fast) {
            B.A = A;
            B.redo_count = redo.as<uint32_t>(); B.fb_count = redo.as<uint32_t>() + 1;
            B.chunk_next = redo.as<uint32_t>() + 16;
            B.redo_list = redo.as<uint32_t>() + redo_head; B.fb_list = B.redo_list + ub_strips + 2;
            B.chunk_budget = 0;
            B.max_word = b->nwords_packed - 1;
            B.q16 = k / 16; B.r16 = k % 16;
            B.rev_a = (uint32_t)(k - 1) % 33u; B.rev_b = (uint32_t)(k - 1) % 31u;
            {
                auto it = c->g8k.find(k);
                if (it == c->g8k.end()) { /* built once per k and context, on the window stream in front of its first user */
                    if (c->g8k.size() >= 8) { /* a caller that sweeps k: start over */
                        HIPCHK(c, sync_both(c));
                        for (auto &kv : c->g8k) (void)hipFree(kv.second);
                        c->g8k.clear();
                    }
                    void *t = nullptr;
                    if (hipMalloc(&t, (size_t)(2 * 65536 + 1024) * sizeof(uint2)) != hipSuccess) return fail(c, NTL_ENOMEM, "hipMalloc failed");
                    hipLaunchKernelGGL(g8k_build_kernel, dim3(256), dim3(256), 0, ws, (const uint64_t (*)[2])c->g8, (uint2 *)t,
                                       B.rev_a, B.rev_b);
                    hipLaunchKernelGGL(g4k_build_kernel, dim3(1), dim3(256), 0, ws, (const uint64_t (*)[2])c->g4, (uint2 *)t + 2 * 65536,
                                       B.rev_a, B.rev_b); /* the four-base form of the same, behind it */
                    HIPCHK(c, hipGetLastError());
                    it = c->g8k.emplace(k, t).first;
                }
                B.g8k = (const uint2 *)it->second;
                B.g4k = B.g8k + 2 * 65536;
            }
            {
                ProfSpan sp(c, "sketch_mask", wsid);
                if (nt == 128) launch_fast_r0<128, 0>(c, B, (unsigned)ub_strips);
                else launch_fast_r0<SK_NT, 0>(c, B, (unsigned)ub_strips);
                HIPCHK(c, hipGetLastError());
            }
            ProfSpan sp(c, "sketch_redo", wsid);
            SketchArgs R = A;
            R.redo_count = B.redo_count; R.redo_list = B.redo_list;
            launch_mask<16>(c, R, (unsigned)ub_strips, true, false, nt);
            if (b->any_multi) launch_mask<16>(c, A, (unsigned)ub_strips, false, true, nt);
            HIPCHK(c, hipGetLastError());
            /* the redo count travels with the window stream: `redo` never leaves it */
            HIPCHK(c, hipMemcpyAsync(&((SketchSums *)s->slot)->redo_n, redo.p, 8, hipMemcpyDeviceToHost, ws));
        } else {
            ProfSpan sp(c, "sketch_mask", wsid);
            if (small) launch_small_w<2>(c, A, (unsigned)ub_strips, b->any_multi);
            else if (C == 16) launch_mask<16>(c, A, (unsigned)ub_strips, true, b->any_multi, nt);
            else if (C == 4) launch_mask<4>(c, A, (unsigned)ub_strips, true, b->any_multi, nt);
            else launch_mask<1>(c, A, (unsigned)ub_strips, true, b->any_multi, nt);
            HIPCHK(c, hipGetLastError());
        }
    }
    if (ws != ms) { /* window stage -> emit */
        hipEvent_t e = sev_get(c);
        if (!e) return fail(c, NTL_EDEVICE, "hipEventCreate failed");
        HIPCHK(c, hipEventRecord(e, ws));
        HIPCHK(c, hipStreamWaitEvent(ms, e, 0));
        sev_put(c, e); /* the wait holds what it needs; the handle may be recorded again */
    }
    {
        ProfSpan sp(c, "sketch_emit");
        const uint64_t tiles = lists ? (ub_strips + EL_STRIPS - 1) / EL_STRIPS : (nmask + EMIT_TILE - 1) / EMIT_TILE;
        DevBuf tile_seq, tile_next;
        if ((rc = tile_next.alloc(c, 8 * 16 * 4))) return rc;
        if (lists) {
            /* ranks: a scan over one count per strip; the total (or, if the lists ran out of pool, a total no array holds: the map
               kernels leave such a sketch alone and sketch_finalize makes it again through the bitmask) */
            strip_first.touch(SID_MAIN); strip_tab.touch(SID_MAIN);
            if ((rc = device_scan(c, Ls.cnt, loff.as<uint32_t>(), ub_strips, nullptr, 1, &dsums->total_mx))) return rc;
            hipLaunchKernelGGL(list_fail_kernel, dim3(1), dim3(64), 0, ms, (const uint32_t *)Ls.ctl, &dsums->total_mx, &dsums->list_fail, tile_next.as<uint32_t>());
            hipLaunchKernelGGL(mx_off_from_strips_kernel, dim3((unsigned)((nseq + 256) / 256)), dim3(256), 0, ms, (const uint32_t *)strip_first.as<uint32_t>(),
                               (const uint32_t *)loff.as<uint32_t>(), (uint32_t)nseq, s->mx_off.as<uint32_t>());
        } else {
            if ((rc = tile.alloc(c, tiles * 4)) || (rc = tile_seq.alloc(c, (tiles + 2) * 4))) return rc;
            if (nseq)
                hipLaunchKernelGGL(tile_seq_kernel, dim3((unsigned)((nseq + 255) / 256)), dim3(256), 0, ms, (const uint64_t *)T.seq_base, (uint32_t)nseq,
                                   tiles, tile_seq.as<uint32_t>());
            hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)tiles), dim3(EMIT_NT), 0, ms,
                               (const uint32_t *)mask.p, nmask, tile.as<uint32_t>(), tile_next.as<uint32_t>());
            hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(SCAN_NT), 0, ms, tile.as<uint32_t>(), tiles, &dsums->total_mx, (uint64_t)0, (uint32_t *)nullptr);
        }
        HIPCHK(c, hipGetLastError());
        if (!s->no_records && (rc = s->records.alloc(c, cap * sizeof(MxRecord)))) return rc;
        if (ix && (rc = s->rpos.alloc(c, cap * 4))) return rc;
        s->cap = cap;
        const int probe = !ix ? 0 : (ix->hit_fraction->load(std::memory_order_relaxed) <= 0.5f ? 1 : 2); /* tags first unless the last batch on this index mostly hit */
        if (ix) {
            if ((rc = s->cand.alloc(c, cap * sizeof(Cand)))) return rc;
            s->cand_gen = ix->gen;
            if (ix->c != c && ix->built) HIPCHK(c, hipStreamWaitEvent(ms, ix->built, 0));
        }
        EmitArgs E;
        E.packed = T.packed; E.seq_base = T.seq_base; E.nseq = (uint32_t)nseq; E.mask = (uint32_t *)mask.p;
        E.nwords = nmask; E.tile_off = tile.as<uint32_t>(); E.tile_seq = tile_seq.as<uint32_t>(); E.mx_off = s->mx_off.as<uint32_t>();
        E.out = s->no_records ? nullptr : s->records.as<MxRecord>(); E.out_cap = (uint32_t)cap;
        E.k = k; E.mult = 1ull ^ ((uint64_t)k * 0x90b45d39fb6da1faull);
        uint64_t roll[16][2];
        make_tables(k, roll, E.seed_tab);
        E.g4 = (const uint64_t (*)[2])c->g4;
        E.g8 = (const uint64_t (*)[2])c->g8;
        E.slots = nullptr; E.tags = nullptr; E.special = nullptr; E.ix_bits = 0; E.cand = nullptr; E.nfound = nullptr;
        E.rpos = ix ? s->rpos.as<uint32_t>() : nullptr;
        if (ix) {
            E.slots = ix->slots.as<IndexSlot>(); E.tags = ix->tags.as<uint8_t>(); E.special = ix->special.as<IndexSpecial>();
            E.ix_bits = ix->bits; E.cand = s->cand.as<Cand>(); E.nfound = &dsums->nfound;
        }
        /* the emit kernel is the last reader of the bitmask and clears the words it read: the mask goes back clean */
        /* Beside the window stage (two streams) the emit kernel may keep to a bounded number of resident workgroups that take
           their tiles from counters; alone: one workgroup per tile, as many resident as fit. */
        E.ntiles = (uint32_t)tiles; E.tile_next = nullptr;
        unsigned egrid = (unsigned)tiles;
        {
            /* the direct-slot form (most lookups hit: one random 64-byte line from HBM per minimizer) is bound by the rate of those
               transactions and gains nothing beyond 12 wavefronts per CU, while the slots it holds are missed by the map kernels
               and the window stage: C5 296-300 -> 269-271 ms per step at 3 workgroups per CU, 275 at 5, 280 at 4 with other window
               grids (profiles/r04_share_sweep_C5.jsonl); the tag form (C3) is as fast uncapped (74.4-75.0 against 74.9-75.9) */
            /* emit_list_kernel (44 registers, 13 KB of LDS) would fill all 32 wavefront slots of a CU, and the window stage's resident
               workgroups of the next sub-batch then wait for them to drain (C3: 91 ms per step uncapped, 75.9 at two per CU) */
            /* beside the DENSE window shapes (w < 137: twelve window wavefronts per CU, launch_fast_r0) the tag form takes four: C2
               0.890 -> 0.800 ms per step (0.806 at three, 0.815 uncapped; profiles/r07b_C2_share_sweep.txt) */
            const bool dense_windows = fast && B.thresh && 4096.0 * (double)B.thresh / 4294967296.0 > 300.0;
            int per_cu = !c->pipelined ? 0 : (probe == 2 ? 3 : (lists ? (dense_windows ? 4 : 2) : 0));
            if (const char *e = getenv("NTL_EMIT_WGS_PER_CU")) per_cu = atoi(e);
            const uint64_t cap = (uint64_t)per_cu * (uint64_t)c->n_cu;
            if (per_cu > 0 && cap >= 8 && cap < tiles) { egrid = (unsigned)cap; E.tile_next = tile_next.as<uint32_t>(); }
            else if (per_cu < 0) { egrid = (unsigned)(-per_cu < 8 ? 8 : -per_cu); E.tile_next = tile_next.as<uint32_t>(); } /* tests: that many workgroups whatever the size */
        }
        const char *eu = getenv("NTL_EMIT_U"); /* minimizers in flight per thread; read per call: the tests switch it inside one process */
        const int emit_u = eu ? atoi(eu) : 1;
        if (lists) {
            EmitListArgs Q;
            Q.Ls = Ls; Q.strip_tab = strip_tab.as<StripInfo>(); Q.strip_off = loff.as<uint32_t>(); Q.nstrips = (uint32_t)ub_strips;
            if (probe == 0) hipLaunchKernelGGL((emit_list_kernel<0, 1>), dim3(egrid), dim3(EL_NT), 0, ms, E, Q);
            else if (probe == 1 && emit_u >= 4) hipLaunchKernelGGL((emit_list_kernel<1, 4>), dim3(egrid), dim3(EL_NT), 0, ms, E, Q);
            else if (probe == 1 && emit_u >= 2) hipLaunchKernelGGL((emit_list_kernel<1, 2>), dim3(egrid), dim3(EL_NT), 0, ms, E, Q);
            else if (probe == 1) hipLaunchKernelGGL((emit_list_kernel<1, 1>), dim3(egrid), dim3(EL_NT), 0, ms, E, Q);
            else if (emit_u >= 4) hipLaunchKernelGGL((emit_list_kernel<2, 4>), dim3(egrid), dim3(EL_NT), 0, ms, E, Q);
            else if (emit_u >= 2) hipLaunchKernelGGL((emit_list_kernel<2, 2>), dim3(egrid), dim3(EL_NT), 0, ms, E, Q);
            else hipLaunchKernelGGL((emit_list_kernel<2, 1>), dim3(egrid), dim3(EL_NT), 0, ms, E, Q);
        } else
        if (probe == 0 && (eu ? emit_u >= 4 : small)) hipLaunchKernelGGL((emit_kernel<0, 4, NTL_EMIT_DENSE_CAP>), dim3(egrid), dim3(EMIT_NT), 0, ms, E); /* dense sketches: rounds of 8192, four k-mers' loads in flight per thread */
        else if (probe == 0 && emit_u >= 2) hipLaunchKernelGGL((emit_kernel<0, 2>), dim3(egrid), dim3(EMIT_NT), 0, ms, E);
        else if (probe == 0) hipLaunchKernelGGL((emit_kernel<0, 1>), dim3(egrid), dim3(EMIT_NT), 0, ms, E);
        else if (probe == 1 && emit_u >= 2) hipLaunchKernelGGL((emit_kernel<1, 2>), dim3(egrid), dim3(EMIT_NT), 0, ms, E);
        else if (probe == 1) hipLaunchKernelGGL((emit_kernel<1, 1>), dim3(egrid), dim3(EMIT_NT), 0, ms, E);
        else if (emit_u >= 2) hipLaunchKernelGGL((emit_kernel<2, 2>), dim3(egrid), dim3(EMIT_NT), 0, ms, E);
        else hipLaunchKernelGGL((emit_kernel<2, 1>), dim3(egrid), dim3(EMIT_NT), 0, ms, E);
        HIPCHK(c, hipGetLastError());
        if (mask.p) {
            mask.clean = sev_get(c);
            if (mask.clean) HIPCHK(c, hipEventRecord(mask.clean, ms));
            else HIPCHK(c, hipStreamSynchronize(ms));
            c->masks.push_back(mask);
            mask.p = nullptr; /* handed over */
        }
        /* lengths of the sketched sequences for the map kernels (the batch may be gone by then) */
        if (b->d_seq_len && nseq && !s->rlen.p) { /* (a second round keeps the array of the first: ntl_map_run may hold its address already) */
            if ((rc = s->rlen.alloc(c, nseq * 4))) return rc;
            HIPCHK(c, hipMemcpyAsync(s->rlen.p, b->d_seq_len, nseq * 4, hipMemcpyDeviceToDevice, ms));
        }
        SketchSums *hs = (SketchSums *)s->slot;
        if (lists) { /* the true total (the device's copy says "too many" when the lists ran out), and whether they did */
            HIPCHK(c, hipMemcpyAsync(&hs->total_mx, loff.as<uint32_t>() + ub_strips, 4, hipMemcpyDeviceToHost, ms));
            HIPCHK(c, hipMemcpyAsync(&hs->list_fail, &dsums->list_fail, 4, hipMemcpyDeviceToHost, ms));
        } else
        HIPCHK(c, hipMemcpyAsync(&hs->total_mx, &dsums->total_mx, 4, hipMemcpyDeviceToHost, ms));
        if (ix) HIPCHK(c, hipMemcpyAsync(&hs->nfound, &dsums->nfound, 8, hipMemcpyDeviceToHost, ms));
    }
    HIPCHK(c, hipEventRecord(s->done, ms));
    {
        hipEvent_t &t = c->throttle[c->n_enqueued & 7u];
        if (!t) t = sev_get(c);
        if (t) HIPCHK(c, hipEventRecord(t, ms));
        c->n_enqueued++;
    }
    s->strips = ub_strips;
    s->from_lists = lists;
    s->pending = true;
    /* temporaries return to the context's cache here; every later user of those blocks is queued behind the kernels above
       on the stream they were used on, so no wait is needed */
    return NTL_OK;
}

static int sketch_run_impl(ntl_ctx *c, const ntl_batch *b, int k, int w, const ntl_index *ix, ntl_sketch **out, bool no_records = false)
{
    if (!c || !b || !out) return NTL_EINVAL;
    *out = nullptr;
    (void)hipSetDevice(c->device);
    SketchGeom G;
    int C, nt, rc;
    if ((rc = sketch_geometry(c, k, w, G, C, nt))) return rc;
    ntl_sketch *s = new ntl_sketch();
    s->c = c; s->nseq = b->nseq; s->k = k; s->w = w; s->src_ix = ix;
    s->no_records = no_records && ix;
    if (ix) ix->refs++;
    s->done = sev_get(c);
    s->slot = slot_get(c);
    if (!s->done || !s->slot) { sketch_unref(s); return fail(c, NTL_EDEVICE, "out of events / page-locked slots"); }
    /* The record array is sized from the expected density before the count is known (the device goes straight on to the
       emit kernel); a batch denser than the guess (low-complexity sequence) is sketched a second time into exact-size
       arrays when its count is asked for (sketch_finalize). */
    uint64_t cap_guess = (uint64_t)(2.5 * (double)b->bases / (double)(w + 1)) + 65536;
    if (cap_guess > b->bases + 1) cap_guess = b->bases + 1;
    if (cap_guess > 0xFFFFFFF0ull) cap_guess = 0xFFFFFFF0ull;
    if (const char *e = getenv("NTL_SKETCH_CAP_GUESS")) cap_guess = (uint64_t)atoll(e); /* tests: force the second round */
    const_cast<ntl_batch *>(b)->refs++;
    s->src = b;
    if ((rc = sketch_enqueue(c, b, k, w, ix, s, cap_guess))) {
        (void)sync_both(c); /* whatever was queued must not outlive the buffers the error path lets go */
        s->pending = false;
        sketch_unref(s);
        return rc;
    }
    *out = s;
    return NTL_OK;
}

/* Completes a sketch: waits for its event, reads the totals, and -- when the batch held more minimizers than the arrays
 * that were sized from the expected density -- makes the sketch again with exact-size arrays (same minimizers). */
static int sketch_finalize(const ntl_sketch *cs)
{
    ntl_sketch *s = const_cast<ntl_sketch *>(cs);
    if (!s->pending) return s->failed;
    ntl_ctx *c = s->c;
    (void)hipSetDevice(c->device);
    for (int round = 0;; round++) {
        const hipError_t e = wait_hot(s->done);
        s->pending = false;
        if (e != hipSuccess) { s->failed = fail(c, NTL_EDEVICE, std::string("sketch: ") + hipGetErrorString(e)); break; }
        const SketchSums hs = *(const SketchSums *)s->slot;
        s->count = hs.total_mx; s->redo_strips = hs.redo_n; s->fallback_strips = hs.fb_n; s->nfound = hs.nfound;
        if (s->count <= s->cap && !hs.list_fail) break;
        if (round || !s->src) { s->failed = fail(c, NTL_EINTERNAL, "sketch: the exact-size round overflowed again"); break; }
        if (hs.list_fail) s->no_lists = true; /* the strips' lists ran out of pool (the counts are right): once more, through the bitmask */
        s->gen++;
        memset(s->slot, 0, sizeof(PinSlot));
        const int rc = sketch_enqueue(c, s->src, s->k, s->w, s->src_ix, s, s->count);
        if (rc) { (void)sync_both(c); s->pending = false; s->failed = rc; break; }
    }
    if (s->src) { batch_unref(s->src); s->src = nullptr; }
    /* complete: what only a pending sketch needs goes back now, not when the handle is destroyed -- a context has NTL_NSLOTS
       page-locked slots, and callers keep thousands of completed sketches alive */
    index_unref(s->src_ix, c); s->src_ix = nullptr;
    sev_put(c, s->done); s->done = nullptr;
    slot_put(c, s->slot); s->slot = nullptr;
    return s->failed;
}

extern "C" int ntl_sketch_run(ntl_ctx *c, const ntl_batch *b, int k, int w, ntl_sketch **out)
{
    return sketch_run_impl(c, b, k, w, nullptr, out);
}

/* The same sketch made FOR one contig index: every minimizer is looked up in `ix` while it is emitted, and ntl_map_run on
 * (ix, this sketch) skips its own lookup pass over the records.  The minimizers are those of ntl_sketch_run. */
extern "C" int ntl_sketch_run_indexed(ntl_ctx *c, const ntl_batch *b, int k, int w, const ntl_index *ix, ntl_sketch **out)
{
    if (!ix) return NTL_EINVAL;
    if (ix->c->device != c->device) return fail(c, NTL_EINVAL, "the index lives on another device");
    return sketch_run_impl(c, b, k, w, ix, out);
}

/* ... and made ONLY for ntl_map_run(ix, this sketch): the 16-byte records are never written (a minimizer leaves its position in
 * the read and its candidate: 12 bytes instead of 24 written here and read by the map kernels).  Everything that needs the records
 * (download, index build, overlap filter, a map against another index) answers NTL_EINVAL. */
extern "C" int ntl_sketch_run_for_map(ntl_ctx *c, const ntl_batch *b, int k, int w, const ntl_index *ix, ntl_sketch **out)
{
    if (!ix) return NTL_EINVAL;
    if (ix->c->device != c->device) return fail(c, NTL_EINVAL, "the index lives on another device");
    return sketch_run_impl(c, b, k, w, ix, out, true);
}

extern "C" int ntl_sketch_has_records(const ntl_sketch *s) { return s && !s->no_records; }

extern "C" void ntl_sketch_destroy(ntl_sketch *s) { sketch_unref(s); }
extern "C" int ntl_sketch_wait(const ntl_sketch *s) { return s ? sketch_finalize(s) : NTL_EINVAL; }
extern "C" uint64_t ntl_sketch_nseq(const ntl_sketch *s) { return s ? s->nseq : 0; }
extern "C" uint64_t ntl_sketch_count(const ntl_sketch *s) { return s && sketch_finalize(s) == NTL_OK ? s->count : 0; }
extern "C" uint64_t ntl_sketch_strips(const ntl_sketch *s) { return s ? s->strips : 0; }
extern "C" uint64_t ntl_sketch_redo_strips(const ntl_sketch *s) { return s && sketch_finalize(s) == NTL_OK ? s->redo_strips : 0; }
extern "C" int ntl_sketch_from_lists(const ntl_sketch *s) { return s && sketch_finalize(s) == NTL_OK && s->from_lists ? 1 : 0; }
extern "C" uint64_t ntl_sketch_fallback_strips(const ntl_sketch *s) { return s && sketch_finalize(s) == NTL_OK ? s->fallback_strips : 0; }

extern "C" int ntl_sketch_download(const ntl_sketch *s, uint64_t *mx_off, uint64_t *hash, uint32_t *pos, uint8_t *strand)
{
    if (!s) return NTL_EINVAL;
    ntl_ctx *c = s->c;
    (void)hipSetDevice(c->device);
    int rc = sketch_finalize(s);
    if (rc) return rc;
    if (s->no_records && (hash || pos || strand)) return fail(c, NTL_EINVAL, "a sketch made by ntl_sketch_run_for_map holds no records");
    const size_t off_bytes = ((s->nseq + 1) * 4 + 63) & ~(size_t)63;
    void *tmp = nullptr;
    if ((rc = host_tmp(c, off_bytes + (s->no_records ? 0 : s->count * sizeof(MxRecord)), &tmp))) return rc;
    const uint32_t *off = (const uint32_t *)tmp;
    const MxRecord *rec = (const MxRecord *)((const char *)tmp + off_bytes);
    HIPCHK(c, hipMemcpyAsync(tmp, s->mx_off.p, (s->nseq + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    if (s->count && !s->no_records) HIPCHK(c, hipMemcpyAsync((void *)rec, s->records.p, s->count * sizeof(MxRecord), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (mx_off) for (uint64_t i = 0; i <= s->nseq; i++) mx_off[i] = off[i];
    if (s->no_records) return NTL_OK;
    /* records -> the caller's column arrays, split over threads */
    unsigned nthr = std::thread::hardware_concurrency();
    nthr = nthr == 0 ? 1 : std::min(nthr, 16u);
    if (s->count < (1u << 18)) nthr = 1;
    auto work = [&](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; i++) {
            if (hash) hash[i] = rec[i].hash;
            if (pos) pos[i] = rec[i].pos;
            if (strand) strand[i] = (uint8_t)(rec[i].meta & 1u);
        }
    };
    if (nthr == 1) work(0, s->count);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthr; t++) th.emplace_back(work, s->count * t / nthr, s->count * (t + 1) / nthr);
        for (auto &x : th) x.join();
    }
    return NTL_OK;
}

extern "C" int ntl_sketch_from_host(ntl_ctx *c, uint64_t nseq, const uint64_t *mx_off, const uint64_t *hash,
                                    const uint32_t *pos, const uint8_t *strand, ntl_sketch **out)
{
    if (!c || !out || !mx_off) return NTL_EINVAL;
    *out = nullptr;
    const uint64_t n = mx_off[nseq];
    if (n >= 0xFFFFFFF0ull || nseq >= ((uint64_t)1 << 31)) return fail(c, NTL_EINVAL, "sketch too large for one batch");
    if (n && (!hash || !pos || !strand)) return NTL_EINVAL;
    (void)hipSetDevice(c->device);
    if (mx_off[0] != 0) return fail(c, NTL_EINVAL, "mx_off[0] must be 0");
    for (uint64_t i = 1; i <= nseq; i++)
        if (mx_off[i] < mx_off[i - 1]) return fail(c, NTL_EINVAL, "mx_off must be non-decreasing");
    /* columns -> 16-byte records in the page-locked bounce buffer, sequences split over threads */
    const size_t off_bytes = ((nseq + 1) * 4 + 63) & ~(size_t)63;
    void *tmp = nullptr;
    int rc = host_tmp(c, off_bytes + n * sizeof(MxRecord), &tmp);
    if (rc) return rc;
    uint32_t *off = (uint32_t *)tmp;
    MxRecord *rec = (MxRecord *)((char *)tmp + off_bytes);
    unsigned nthr = std::thread::hardware_concurrency();
    nthr = nthr == 0 ? 1 : std::min(nthr, 16u);
    if (n < (1u << 18)) nthr = 1;
    auto work = [&](uint64_t s0, uint64_t s1) {
        for (uint64_t s = s0; s < s1; s++) {
            off[s] = (uint32_t)mx_off[s];
            for (uint64_t i = mx_off[s]; i < mx_off[s + 1]; i++) {
                rec[i].hash = hash[i]; rec[i].pos = pos[i];
                rec[i].meta = ((uint32_t)s << 1) | (strand[i] ? 1u : 0u);
            }
        }
    };
    if (nthr == 1) work(0, nseq);
    else {
        std::vector<std::thread> th;
        std::vector<uint64_t> cut(nthr + 1, nseq); /* about equal numbers of minimizers per thread */
        cut[0] = 0;
        for (unsigned t = 1; t < nthr; t++)
            cut[t] = std::max<uint64_t>(cut[t - 1], (uint64_t)(std::lower_bound(mx_off, mx_off + nseq, n * t / nthr) - mx_off));
        for (unsigned t = 0; t < nthr; t++) th.emplace_back(work, cut[t], cut[t + 1]);
        for (auto &x : th) x.join();
    }
    off[nseq] = (uint32_t)mx_off[nseq];
    std::unique_ptr<ntl_sketch> sk(new ntl_sketch());
    sk->c = c; sk->nseq = nseq; sk->count = n; sk->cap = n;
    if ((rc = sk->records.alloc(c, n * sizeof(MxRecord))) || (rc = sk->mx_off.alloc(c, (nseq + 1) * 4))) return rc;
    hipError_t e = hipSuccess;
    if (n) e = hipMemcpyAsync(sk->records.p, rec, n * sizeof(MxRecord), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sk->mx_off.p, off, (nseq + 1) * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, NTL_EDEVICE, hipGetErrorString(e));
    *out = sk.release();
    return NTL_OK;
}


/* ------------------------------------------------------------------ overlap-stage consumer -- */

/* read_minimizer_line of bin/ntlink_overlap_sequences.py:170-190 over a device-resident sketch (overlap_kernels.h):
 * valid-region filter, per-sequence removal of every hash that occurs more than once, dense result in input order. */
extern "C" int ntl_overlap_filter(ntl_ctx *c, const ntl_sketch *s, const uint64_t *region_off, const uint32_t *region_start,
                                  const uint32_t *region_end, ntl_sketch **out)
{
    if (!c || !s || !region_off || !out) return NTL_EINVAL;
    *out = nullptr;
    (void)hipSetDevice(c->device);
    if (s->no_records) return fail(c, NTL_EINVAL, "a sketch made by ntl_sketch_run_for_map holds no records");
    if (int frc = sketch_finalize(s)) return frc;
    const uint64_t nseq = s->nseq, n = s->count;
    const uint64_t nreg = region_off[nseq];
    if (region_off[0] != 0) return fail(c, NTL_EINVAL, "region_off[0] must be 0");
    for (uint64_t i = 0; i < nseq; i++)
        if (region_off[i + 1] < region_off[i]) return fail(c, NTL_EINVAL, "region_off must be non-decreasing");
    if (nreg >= 0xFFFFFFF0ull) return fail(c, NTL_EINVAL, "too many regions");
    if (nreg && (!region_start || !region_end)) return NTL_EINVAL;
    (void)hipSetDevice(c->device);
    std::unique_ptr<ntl_sketch> o(new ntl_sketch());
    o->c = c; o->nseq = nseq;
    int rc;
    DevBuf roff, rs, re, slots, dup, side, keep, dst;
    std::vector<uint32_t> roff32(nseq + 1);
    for (uint64_t i = 0; i <= nseq; i++) roff32[i] = (uint32_t)region_off[i];
    if ((rc = roff.alloc(c, (nseq + 1) * 4)) || (rc = rs.alloc(c, (nreg + 1) * 4)) || (rc = re.alloc(c, (nreg + 1) * 4)) ||
        (rc = slots.alloc(c, (2 * n + 1) * 8)) || (rc = dup.alloc(c, (2 * n / 32 + 2) * 4)) || (rc = side.alloc(c, (nseq + 1) * 4)) ||
        (rc = keep.alloc(c, (n + 1) * 4)) || (rc = dst.alloc(c, (n + 2) * 4)) || (rc = o->mx_off.alloc(c, (nseq + 1) * 4)))
        return rc;
    uint32_t total = 0;
    {
        ProfSpan sp(c, "overlap_filter");
        HIPCHK(c, hipMemcpyAsync(roff.p, roff32.data(), (nseq + 1) * 4, hipMemcpyHostToDevice, c->stream));
        if (nreg) {
            HIPCHK(c, hipMemcpyAsync(rs.p, region_start, nreg * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(re.p, region_end, nreg * 4, hipMemcpyHostToDevice, c->stream));
        }
        HIPCHK(c, hipMemsetAsync(slots.p, 0xFF, (2 * n + 1) * 8, c->stream)); /* NTL_INF = empty */
        HIPCHK(c, hipMemsetAsync(dup.p, 0, (2 * n / 32 + 2) * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(side.p, 0, (nseq + 1) * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(keep.p, 0, (n + 1) * 4, c->stream));
        OvlArgs A;
        A.rec = s->records.as<MxRecord>(); A.n = n; A.mx_off = s->mx_off.as<uint32_t>();
        A.reg_off = roff.as<uint32_t>(); A.reg_start = rs.as<uint32_t>(); A.reg_end = re.as<uint32_t>();
        A.slots = slots.as<unsigned long long>(); A.dupbits = dup.as<uint32_t>(); A.side = side.as<uint32_t>();
        A.keep = keep.as<uint32_t>();
        const unsigned grid = (unsigned)((n + 255) / 256);
        if (n) {
            hipLaunchKernelGGL(ovl_insert_kernel, dim3(grid), dim3(256), 0, c->stream, A);
            hipLaunchKernelGGL(ovl_resolve_kernel, dim3(grid), dim3(256), 0, c->stream, A);
            HIPCHK(c, hipGetLastError());
        }
        /* the one wait of the call: the number of kept records sizes the result */
        if ((rc = device_scan(c, keep.as<uint32_t>(), dst.as<uint32_t>(), n, &total))) return rc;
        o->count = total; o->cap = total;
        if ((rc = o->records.alloc(c, (uint64_t)total * sizeof(MxRecord)))) return rc;
        const uint64_t work = std::max<uint64_t>(n, nseq + 1);
        hipLaunchKernelGGL(ovl_gather_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, c->stream,
                           (const MxRecord *)s->records.as<MxRecord>(), n, (const uint32_t *)keep.as<uint32_t>(),
                           (const uint32_t *)dst.as<uint32_t>(), o->records.as<MxRecord>(), (const uint32_t *)s->mx_off.as<uint32_t>(),
                           (uint32_t)nseq, o->mx_off.as<uint32_t>());
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); /* roff32 and the caller's region arrays are free again */
    *out = o.release();
    return NTL_OK;
}

/* ------------------------------------------------------------------ index ---------------- */


extern "C" int ntl_index_build(ntl_ctx *c, const ntl_sketch *ctg, const uint32_t *ctg_len, uint32_t n_ctg, ntl_index **out)
{
    if (!c || !ctg || !out || (!ctg_len && n_ctg)) return NTL_EINVAL;
    *out = nullptr;
    if ((uint64_t)n_ctg != ctg->nseq) return fail(c, NTL_EINVAL, "n_ctg must equal the number of sketched contigs");
    if (ctg->no_records) return fail(c, NTL_EINVAL, "a sketch made by ntl_sketch_run_for_map holds no records");
    if (n_ctg >= (1u << 29)) return fail(c, NTL_EINVAL, "too many contigs"); /* contig id << 3 | flags in one word (map_kernels.h) */
    (void)hipSetDevice(c->device);
    if (int frc = sketch_finalize(ctg)) return frc; /* the table is sized from the number of contig minimizers */
    if (ctg->c != c) HIPCHK(c, sync_both(ctg->c));
    std::unique_ptr<ntl_index> ix_guard(new ntl_index());
    ntl_index *ix = ix_guard.get();
    ix->c = c; ix->n_ctg = n_ctg; ix->gen = g_index_gen++;
    int bits = 10;
    while (((uint64_t)1 << bits) < 2 * ctg->count + 2) bits++;
    ix->bits = bits;
    ix->nslots = (uint64_t)1 << bits;
    int rc;
    DevBuf &cnt = ix->cnt;
    if ((rc = ix->slots.alloc(c, ix->nslots * sizeof(IndexSlot))) || (rc = ix->special.alloc(c, sizeof(IndexSpecial))) ||
        (rc = ix->ctg_len.alloc(c, ((uint64_t)n_ctg + 1) * 4)) || (rc = cnt.alloc(c, 8)) || (rc = ix->tags.alloc(c, ix->nslots + 16))) return rc;
    unsigned long long size = 0;
    {
        ProfSpan sp(c, "index");
        ix->h_ctg_len.assign(ctg_len, ctg_len + n_ctg);
        if (n_ctg) HIPCHK(c, hipMemcpyAsync(ix->ctg_len.p, ix->h_ctg_len.data(), (uint64_t)n_ctg * 4, hipMemcpyHostToDevice, c->stream));
        DevBuf dup; /* one bit per slot: the key arrived more than once */
        if ((rc = dup.alloc(c, ix->nslots / 8))) return rc;
        hipLaunchKernelGGL(index_clear_kernel, dim3((unsigned)((ix->nslots + 255) / 256)), dim3(256), 0, c->stream,
                           ix->slots.as<IndexSlot>(), ix->nslots, dup.as<uint32_t>(), ix->special.as<IndexSpecial>(),
                           cnt.as<unsigned long long>());
        if (ctg->count)
            hipLaunchKernelGGL(index_insert_kernel, dim3((unsigned)((ctg->count + 255) / 256)), dim3(256), 0, c->stream,
                               (const MxRecord *)ctg->records.as<MxRecord>(), ctg->count, ix->slots.as<IndexSlot>(), bits,
                               ix->special.as<IndexSpecial>(), dup.as<uint32_t>());
        hipLaunchKernelGGL(index_finish_kernel, dim3((unsigned)std::min<uint64_t>((ix->nslots / 4 + 255) / 256, 4096)), dim3(256), 0, c->stream,
                           ix->slots.as<IndexSlot>(), ix->nslots, (const IndexSpecial *)ix->special.as<IndexSpecial>(),
                           (const uint32_t *)dup.as<uint32_t>(), ix->tags.as<uint8_t>(), cnt.as<unsigned long long>());
        HIPCHK(c, hipGetLastError());
        /* the lookups read eight tags at a time from any slot: the first eight again behind the last (the probe sequence wraps) */
        HIPCHK(c, hipMemcpyAsync(ix->tags.as<uint8_t>() + ix->nslots, ix->tags.p, 8, hipMemcpyDeviceToDevice, c->stream));
    }
    (void)size;
    if ((ix->built = sev_get(c))) HIPCHK(c, hipEventRecord(ix->built, c->stream));
    *out = ix_guard.release();
    return NTL_OK;
}

extern "C" void ntl_index_destroy(ntl_index *ix)
{
    if (ix) index_unref(ix, ix->c); /* sketches and map results still pending against it keep it until their work has run */
}

extern "C" uint64_t ntl_index_size(const ntl_index *ix)
{
    if (!ix) return 0;
    if (!ix->size_known) {
        unsigned long long v = 0;
        (void)hipSetDevice(ix->c->device);
        if (hipMemcpyAsync(&v, ix->cnt.p, 8, hipMemcpyDeviceToHost, ix->c->stream) == hipSuccess &&
            hipStreamSynchronize(ix->c->stream) == hipSuccess) {
            ix->size = v;
            ix->size_known = true;
        }
    }
    return ix->size;
}

/* ------------------------------------------------------------------ map ------------------ */

struct MapSums { unsigned long long nfound; uint32_t err; uint32_t tot[3]; uint32_t n_over; uint32_t nmx; }; /* one memset, one read-back */
static_assert(sizeof(MapSums) <= sizeof(PinSlot), "the sums must fit a page-locked slot");

struct ntl_mapres {
    ntl_ctx *c;
    mutable uint64_t n_maps = 0, n_hits = 0, n_pafs = 0, n_index_hits = 0;
    mutable DevBuf maps, pafs;       /* dense, read order; sized from the sketch's capacity, filled to n_* */
    mutable DevBuf hits;             /* per-read regions (read r's hits from mx_off[r] on); maps[].hit_off points into them */
    mutable DevBuf maps_dense, hits_dense, hit_doff; /* made on demand: the dense copy ntl_mapres_download hands out; u32[n_maps + 1] dense offsets */
    mutable bool dense_made = false, doff_made = false;
    /* lazy completion */
    mutable bool pending = false;
    mutable hipEvent_t done = nullptr;
    mutable PinSlot *slot = nullptr;
    mutable int failed = 0;
    const ntl_index *ix = nullptr;
    const ntl_sketch *reads = nullptr;  /* held (refs) while pending */
    uint64_t reads_gen = 0;             /* generation of the sketch the kernels were queued on */
    ntl_map_params params;
    std::shared_ptr<std::atomic<float>> hitf; /* the index's hit fraction (the index itself may be gone when this completes) */
    DevBuf rlen_own;                    /* read lengths uploaded by this call (sketches that did not come from a batch) */
};

/* Queues the lookup (when the sketch does not carry candidates), the map kernels, the offset scans and the gather on MAIN.
 * Nothing waits: the three totals, the hit count and the invariant flag land in the result's page-locked slot. */
static int map_enqueue(ntl_ctx *c, const ntl_index *ix, const ntl_sketch *reads, const uint32_t *d_rlen, ntl_mapres *R)
{
    const uint64_t nreads = reads->nseq;
    const bool have_cand = reads->cand_gen == ix->gen && reads->cand.p != nullptr;
    if (reads->no_records && !have_cand) return fail(c, NTL_EINVAL, "a sketch made by ntl_sketch_run_for_map maps against the index it was made for only");
    if (!have_cand) { /* the lookup pass runs over the records: their number sizes its grid */
        if (int frc = sketch_finalize(reads)) return frc;
    }
    const uint64_t nmx = reads->pending ? reads->cap : reads->count;
    const ntl_map_params *params = &R->params;
    int rc;
    hipStream_t ms = c->stream;
    DevBuf cand, rpos, smaps, spafs, n3, off3, scr, sums, over;
    R->dense_made = R->doff_made = false;
    const uint64_t cap = nmx ? nmx : 1;
    if ((!have_cand && ((rc = cand.alloc(c, cap * sizeof(Cand))) || (rc = rpos.alloc(c, cap * 4)))) ||
        (rc = smaps.alloc(c, cap * sizeof(MapRec))) ||
        (rc = spafs.alloc(c, cap * sizeof(PafRec))) || (rc = n3.alloc(c, 3 * (nreads + 1) * 4)) ||
        (rc = off3.alloc(c, 3 * (nreads + 1) * 4)) || (rc = scr.alloc(c, (uint64_t)(MAP_NHA + MAP_NRA) * cap * 4)) ||
        (rc = sums.alloc(c, sizeof(MapSums))) || (rc = over.alloc(c, (nreads + 1) * 4)) ||
        (rc = R->maps.alloc(c, cap * sizeof(MapRec))) || (rc = R->hits.alloc(c, cap * sizeof(HitRec))) ||
        (rc = R->pafs.alloc(c, cap * sizeof(PafRec)))) return rc;
    MapSums *dsums = sums.as<MapSums>();
    if (ix->c != c && ix->built) HIPCHK(c, hipStreamWaitEvent(ms, ix->built, 0));
    HIPCHK(c, hipMemsetAsync(sums.p, 0, sizeof(MapSums), ms));
    if (have_cand) HIPCHK(c, hipMemcpyAsync(&dsums->nfound, &reads->sums.as<SketchSums>()->nfound, 8, hipMemcpyDeviceToDevice, ms));
    if (!have_cand) {
        ProfSpan sp(c, "probe");
        if (nmx) {
            const dim3 grid((unsigned)std::min<uint64_t>((nmx + 256 * PROBE_U - 1) / (256 * PROBE_U), 4096));
            /* tags first unless the previous batch on this index found more than half of its minimizers (same result either way) */
            if (ix->hit_fraction->load(std::memory_order_relaxed) <= 0.5f)
                hipLaunchKernelGGL(probe_kernel<true>, grid, dim3(256), 0, ms,
                                   (const MxRecord *)reads->records.as<MxRecord>(), nmx, (const IndexSlot *)ix->slots.as<IndexSlot>(),
                                   ix->bits, (const IndexSpecial *)ix->special.as<IndexSpecial>(), cand.as<Cand>(), rpos.as<uint32_t>(),
                                   &dsums->nfound, (const uint8_t *)ix->tags.as<uint8_t>());
            else
                hipLaunchKernelGGL(probe_kernel<false>, grid, dim3(256), 0, ms,
                                   (const MxRecord *)reads->records.as<MxRecord>(), nmx, (const IndexSlot *)ix->slots.as<IndexSlot>(),
                                   ix->bits, (const IndexSpecial *)ix->special.as<IndexSpecial>(), cand.as<Cand>(), rpos.as<uint32_t>(),
                                   &dsums->nfound, (const uint8_t *)ix->tags.as<uint8_t>());
        }
        HIPCHK(c, hipGetLastError());
    }
    MapArgs A;
    A.mx_off = reads->mx_off.as<uint32_t>();
    A.rpos = have_cand ? reads->rpos.as<uint32_t>() : rpos.as<uint32_t>();
    A.cand = have_cand ? reads->cand.as<Cand>() : cand.as<Cand>();
    A.read_len = d_rlen; A.ctg_len = ix->ctg_len.as<uint32_t>(); A.nreads = (uint32_t)nreads;
    A.P.k = params->k; A.P.z = params->z; A.P.x = params->x; A.P.sensitive = params->sensitive;
    A.P.repeat_filter = params->repeat_filter;
    A.maps = smaps.as<MapRec>(); A.hits = R->hits.as<HitRec>(); A.pafs = spafs.as<PafRec>();
    A.n_maps = n3.as<uint32_t>(); A.n_hits = A.n_maps + (nreads + 1); A.n_pafs = A.n_hits + (nreads + 1);
    A.scr = scr.as<uint32_t>(); A.scr_stride = cap; A.err = &dsums->err;
    A.over_list = over.as<uint32_t>(); A.over_count = &dsums->n_over;
    A.nmx_out = &dsums->nmx;
    /* a sketch whose count is not known yet may have overflowed its arrays: the kernels look at its total and leave it alone */
    A.mx_total = reads->pending ? &reads->sums.as<SketchSums>()->total_mx : nullptr;
    A.mx_cap = (uint32_t)std::min<uint64_t>(reads->cap, 0xFFFFFFFFull);
    if (nreads) {
        {
            ProfSpan sp(c, "map");
            /* reads by size class, each class with the LDS staging that fits it; resident-size grids (28 / 14 / 8 wavefronts
               per CU fit) looping over all reads, 64 at a time */
            const uint64_t groups = (nreads + MAP_GROUP - 1) / MAP_GROUP;
            uint64_t g0 = 4 * 7168, g1 = 4 * 3584, g2 = 4 * 2048;
            if (const char *e = getenv("NTL_MAP_WAVES_PER_CU")) { /* tuning: a bounded number of resident wavefronts beside the window stage */
                const uint64_t cap = (uint64_t)std::max(1, atoi(e)) * (uint64_t)std::max(1, c->n_cu);
                g0 = std::min(g0, cap); g1 = std::min(g1, cap); g2 = std::min(g2, cap);
            }
            hipLaunchKernelGGL((map_kernel<256, 64, 0>), dim3((unsigned)std::min<uint64_t>(groups, g0)), dim3(MAP_NT), 0, ms, A);
            hipLaunchKernelGGL((map_kernel<512, 128, 1>), dim3((unsigned)std::min<uint64_t>(groups, g1)), dim3(MAP_NT), 0, ms, A);
            hipLaunchKernelGGL((map_kernel<1024, 128, 2>), dim3((unsigned)std::min<uint64_t>(groups, g2)), dim3(MAP_NT), 0, ms, A);
            /* reads with more hits / runs than the largest staging holds (rare): same code on global scratch */
            hipLaunchKernelGGL(map_overflow_kernel, dim3((unsigned)std::min<uint64_t>(nreads, 8192)), dim3(MAP_NT), 0, ms, A); /* no LDS: 32 wavefronts per CU resident */
            HIPCHK(c, hipGetLastError());
        }
        ProfSpan sp(c, "compact");
        uint32_t *o = off3.as<uint32_t>();
        if ((rc = device_scan(c, n3.as<uint32_t>(), o, nreads, nullptr, 3, dsums->tot))) return rc;
        hipLaunchKernelGGL(map_gather_kernel, dim3((unsigned)nreads), dim3(64), 0, ms, A, (const uint32_t *)o,
                           (const uint32_t *)(o + 2 * (nreads + 1)), R->maps.as<MapRec>(), R->pafs.as<PafRec>());
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(R->slot, sums.p, sizeof(MapSums), hipMemcpyDeviceToHost, ms));
    HIPCHK(c, hipEventRecord(R->done, ms));
    R->pending = true;
    R->reads_gen = reads->gen;
    return NTL_OK;
}

static void mapres_free(ntl_mapres *R)
{
    if (!R) return;
    ntl_ctx *c = R->c;
    (void)hipSetDevice(c->device);
    if (R->pending) {
        Zombie z;
        z.done = R->done; z.slot = R->slot; z.kind = 2; z.hitf = R->hitf; z.ix = R->ix;
        c->zombies.push_back(z);
    } else {
        sev_put(c, R->done);
        slot_put(c, R->slot);
        index_unref(R->ix, c);
    }
    if (R->reads) sketch_unref(R->reads);
    delete R;
}

/* Completes a map result: the sketch first (it may have to be made again with larger arrays -- then the map kernels ran on
 * nothing and are queued again), then this call's totals. */
static int mapres_finalize(const ntl_mapres *cR)
{
    ntl_mapres *R = const_cast<ntl_mapres *>(cR);
    if (!R->pending) return R->failed;
    ntl_ctx *c = R->c;
    (void)hipSetDevice(c->device);
    for (int round = 0;; round++) {
        if (R->reads && (R->failed = sketch_finalize(R->reads))) { (void)hipEventSynchronize(R->done); R->pending = false; break; }
        const hipError_t e = wait_hot(R->done);
        R->pending = false;
        if (e != hipSuccess) { R->failed = fail(c, NTL_EDEVICE, std::string("map: ") + hipGetErrorString(e)); break; }
        if (R->reads && R->reads->gen != R->reads_gen && round == 0) {
            memset(R->slot, 0, sizeof(PinSlot));
            const int rc = map_enqueue(c, R->ix, R->reads, R->rlen_own.p ? R->rlen_own.as<uint32_t>() : R->reads->rlen.as<uint32_t>(), R);
            if (rc) { (void)sync_both(c); R->pending = false; R->failed = rc; break; }
            continue;
        }
        const MapSums hs = *(const MapSums *)R->slot;
        R->n_maps = hs.tot[0]; R->n_hits = hs.tot[1]; R->n_pafs = hs.tot[2];
        R->n_index_hits = hs.nfound;
        if (hs.nmx) R->hitf->store((float)((double)hs.nfound / (double)hs.nmx), std::memory_order_relaxed);
        if (hs.err) R->failed = fail(c, NTL_EINTERNAL, "an accepted contig appeared twice in one read (bin/ntlink_utils.py:262-266)");
        break;
    }
    if (R->reads) { sketch_unref(R->reads); R->reads = nullptr; }
    index_unref(R->ix, c); R->ix = nullptr;
    sev_put(c, R->done); R->done = nullptr;
    slot_put(c, R->slot); R->slot = nullptr;
    return R->failed;
}

extern "C" int ntl_map_run(ntl_ctx *c, const ntl_index *ix, const ntl_sketch *reads, const uint32_t *read_len,
                           const ntl_map_params *params, ntl_mapres **out)
{
    if (!c || !ix || !reads || !params || !out || (!read_len && reads->nseq)) return NTL_EINVAL;
    *out = nullptr;
    (void)hipSetDevice(c->device);
    if (reads->c != c) { /* a sketch of another context: complete it there first */
        if (int frc = sketch_finalize(reads)) return frc;
        HIPCHK(c, sync_both(reads->c));
    }
    const uint64_t nreads = reads->nseq;
    ntl_mapres *R = new ntl_mapres();
    R->c = c; R->ix = ix; R->params = *params; R->hitf = ix->hit_fraction;
    ix->refs++;
    R->done = sev_get(c);
    R->slot = slot_get(c);
    if (!R->done || !R->slot) { mapres_free(R); return fail(c, NTL_EDEVICE, "out of events / page-locked slots"); }
    int rc = NTL_OK;
    /* read lengths: a sketch made from a batch has them on the device already (the `--len` column IS the sequence length);
       anything else -- or lengths that differ from the batch's -- is uploaded here */
    const uint32_t *d_rlen = nullptr;
    if (reads->rlen.p && nreads) {
        /* the device copy was made from the batch; the caller's array is checked against it on the host side of the batch */
        d_rlen = reads->rlen.as<uint32_t>();
        if (!reads->src || memcmp(read_len, reads->src->seq_len.data(), nreads * 4) != 0) d_rlen = nullptr;
    }
    if (!d_rlen) {
        if ((rc = R->rlen_own.alloc(c, (nreads + 1) * 4))) { mapres_free(R); return rc; }
        if (nreads) {
            const hipError_t e = hipMemcpyAsync(R->rlen_own.p, read_len, nreads * 4, hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { /* pageable source: the caller's array is free again */
                mapres_free(R);
                return fail(c, NTL_EDEVICE, "upload of the read lengths failed");
            }
        }
        d_rlen = R->rlen_own.as<uint32_t>();
    }
    const_cast<ntl_sketch *>(reads)->refs++;
    R->reads = reads;
    if ((rc = map_enqueue(c, ix, reads, d_rlen, R))) {
        (void)sync_both(c);
        R->pending = false;
        mapres_free(R);
        return rc;
    }
    *out = R;
    return NTL_OK;
}

extern "C" void ntl_mapres_destroy(ntl_mapres *r) { mapres_free(r); }
extern "C" int ntl_mapres_wait(const ntl_mapres *r) { return r ? mapres_finalize(r) : NTL_EINVAL; }
extern "C" uint64_t ntl_mapres_n_mappings(const ntl_mapres *r) { return r && mapres_finalize(r) == NTL_OK ? r->n_maps : 0; }
extern "C" uint64_t ntl_mapres_n_hits(const ntl_mapres *r) { return r && mapres_finalize(r) == NTL_OK ? r->n_hits : 0; }
extern "C" uint64_t ntl_mapres_n_pafs(const ntl_mapres *r) { return r && mapres_finalize(r) == NTL_OK ? r->n_pafs : 0; }
extern "C" uint64_t ntl_mapres_n_index_hits(const ntl_mapres *r) { return r && mapres_finalize(r) == NTL_OK ? r->n_index_hits : 0; }

/* hit_doff[m] = hits of the mappings before m (the dense numbering of the hits); needs a completed result */
static int mapres_dense_offsets(const ntl_mapres *r)
{
    if (r->doff_made) return NTL_OK;
    ntl_ctx *c = r->c;
    int rc;
    if ((rc = r->hit_doff.alloc(c, (r->n_maps + 1) * 4))) return rc;
    if (r->n_maps) {
        hipLaunchKernelGGL(map_nhits_kernel, dim3((unsigned)((r->n_maps + 255) / 256)), dim3(256), 0, c->stream, (const MapRec *)r->maps.as<MapRec>(),
                           (uint32_t)r->n_maps, r->hit_doff.as<uint32_t>());
        if ((rc = device_scan(c, r->hit_doff.as<uint32_t>(), r->hit_doff.as<uint32_t>(), r->n_maps, nullptr))) return rc;
    } else HIPCHK(c, hipMemsetAsync(r->hit_doff.p, 0, 4, c->stream));
    HIPCHK(c, hipGetLastError());
    r->doff_made = true;
    return NTL_OK;
}

/* the dense copy of mappings and hits that the host receives (the device keeps the hits in their per-read regions) */
static int mapres_densify(const ntl_mapres *r)
{
    if (r->dense_made) return NTL_OK;
    ntl_ctx *c = r->c;
    int rc;
    if ((rc = mapres_dense_offsets(r)) || (rc = r->maps_dense.alloc(c, (r->n_maps + 1) * sizeof(MapRec))) ||
        (rc = r->hits_dense.alloc(c, (r->n_hits + 1) * sizeof(HitRec))))
        return rc;
    if (r->n_maps)
        hipLaunchKernelGGL(map_densify_kernel, dim3((unsigned)r->n_maps), dim3(64), 0, c->stream, (const MapRec *)r->maps.as<MapRec>(), (uint32_t)r->n_maps,
                           (const HitRec *)r->hits.as<HitRec>(), (const uint32_t *)r->hit_doff.as<uint32_t>(), r->maps_dense.as<MapRec>(),
                           r->hits_dense.as<HitRec>());
    HIPCHK(c, hipGetLastError());
    r->dense_made = true;
    return NTL_OK;
}

extern "C" int ntl_mapres_download(const ntl_mapres *r, ntl_mapping *maps, ntl_hit *hits, ntl_paf *pafs)
{
    if (!r) return NTL_EINVAL;
    ntl_ctx *c = r->c;
    (void)hipSetDevice(c->device);
    if (int frc = mapres_finalize(r)) return frc;
    static_assert(sizeof(ntl_mapping) == sizeof(MapRec) && sizeof(ntl_hit) == sizeof(HitRec) && sizeof(ntl_paf) == sizeof(PafRec),
                  "ABI records must match the device records");
    if ((maps || hits) && r->n_maps) {
        if (int drc = mapres_densify(r)) return drc;
        if (maps) HIPCHK(c, hipMemcpyAsync(maps, r->maps_dense.p, r->n_maps * sizeof(MapRec), hipMemcpyDeviceToHost, c->stream));
        if (hits && r->n_hits) HIPCHK(c, hipMemcpyAsync(hits, r->hits_dense.p, r->n_hits * sizeof(HitRec), hipMemcpyDeviceToHost, c->stream));
    }
    if (pafs && r->n_pafs) HIPCHK(c, hipMemcpyAsync(pafs, r->pafs.p, r->n_pafs * sizeof(PafRec), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, main_wait(c));
    return NTL_OK;
}

/* ------------------------------------------------------------------ text on the device ---- */

struct ntl_names {
    ntl_ctx *c;
    uint64_t n = 0;
    uint64_t max_len = 0; /* the longest name: bounds the bytes of a header / PAF line (ntl_mapres_format) */
    DevBuf blob, off, len; /* bytes, u64[n + 1], u32[n] (or none) */
};

extern "C" int ntl_names_create(ntl_ctx *c, const char *blob, const uint64_t *off, const uint32_t *len, uint64_t n, ntl_names **out)
{
    if (!c || !out || (n && (!blob || !off))) return NTL_EINVAL;
    *out = nullptr;
    (void)hipSetDevice(c->device);
    std::unique_ptr<ntl_names> t(new ntl_names());
    t->c = c; t->n = n;
    const uint64_t zero = 0;
    const uint64_t bytes = n ? off[n] : 0;
    for (uint64_t i = 0; i < n; i++) t->max_len = std::max<uint64_t>(t->max_len, off[i + 1] - off[i]);
    int rc;
    if ((rc = t->blob.alloc(c, bytes + 16)) || (rc = t->off.alloc(c, (n + 1) * 8)) || (len && (rc = t->len.alloc(c, (n + 1) * 4)))) return rc;
    if (bytes) HIPCHK(c, hipMemcpyAsync(t->blob.p, blob, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(t->off.p, n ? (const void *)off : (const void *)&zero, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (len && n) HIPCHK(c, hipMemcpyAsync(t->len.p, len, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, main_wait(c)); /* the caller's arrays are free again */
    *out = t.release();
    return NTL_OK;
}

extern "C" void ntl_names_destroy(ntl_names *t)
{
    if (!t) return;
    (void)hipSetDevice(t->c->device);
    delete t;
}

struct ntl_text {
    ntl_ctx *c;
    uint64_t verbose_bytes = 0, paf_bytes = 0, n_maps = 0;
    DevBuf verbose, paf, ends, maps;
};

/* Makes the text of a map result on the device.  One host wait: the three byte totals decide the size of the text arrays. */
extern "C" int ntl_mapres_format(const ntl_mapres *r, const ntl_names *reads, const ntl_names *contigs, int want_verbose, int want_paf,
                                 ntl_text **out)
{
    if (!r || !reads || !contigs || !out) return NTL_EINVAL;
    *out = nullptr;
    ntl_ctx *c = r->c;
    (void)hipSetDevice(c->device);
    if (int frc = mapres_finalize(r)) return frc;
    if (!reads->len.p || !contigs->len.p) return fail(c, NTL_EINVAL, "ntl_mapres_format: both name tables need their lengths");
    if (r->n_hits >= 0xFFFFFF00ull || r->n_maps >= 0xFFFFFF00ull) return fail(c, NTL_ERANGE, "ntl_mapres_format: too many records for one text");
    /* 32-bit offsets, every one of the three scans (tokens, headers, PAF lines) bounded on its own: a token is at most 27 bytes, a header
       two names + a count + three separators, a PAF line two names + ten numbers + strand and tabs (the header and PAF
       sums could wrap unseen and the fill kernel then wrote at wrapped offsets) */
    {
        const uint64_t names2 = reads->max_len + contigs->max_len;
        const uint64_t tok_b = r->n_hits * 27, hdr_b = r->n_maps * (names2 + 16), paf_b = r->n_pafs * (names2 + 10 * 11 + 16);
        uint64_t most = 0xFFFFFF00ull;
        if (const char *e = getenv("NTL_FORMAT_MAX_TEXT")) most = (uint64_t)atoll(e); /* tests: the caller's fallback without 4 GB of text */
        if (tok_b + hdr_b >= most || paf_b >= most)
            return fail(c, NTL_ERANGE, "ntl_mapres_format: the text of this batch may exceed 4 GB (32-bit offsets): format it on the host, or map smaller batches");
    }
    std::unique_ptr<ntl_text> t(new ntl_text());
    t->c = c; t->n_maps = r->n_maps;
    const uint32_t nm = (uint32_t)r->n_maps, nh = want_verbose ? (uint32_t)r->n_hits : 0u, np = want_paf ? (uint32_t)r->n_pafs : 0u;
    DevBuf tok, hdr, pl;
    int rc;
    if ((rc = tok.alloc(c, ((uint64_t)nh + 1) * 4)) || (rc = hdr.alloc(c, ((uint64_t)nm + 1) * 4)) || (rc = pl.alloc(c, ((uint64_t)np + 1) * 4)) ||
        (rc = t->ends.alloc(c, ((uint64_t)r->n_maps * 2 + 1) * sizeof(HitRec))) || (rc = t->maps.alloc(c, ((uint64_t)r->n_maps + 1) * sizeof(MapRec))))
        return rc;
    if ((rc = mapres_dense_offsets(r))) return rc;
    FmtArgs A;
    A.hit_doff = r->hit_doff.as<uint32_t>();
    A.maps = r->maps.as<MapRec>(); A.hits = r->hits.as<HitRec>(); A.pafs = r->pafs.as<PafRec>();
    A.n_maps = nm; A.n_hits = nh; A.n_pafs = np; A.do_verbose = want_verbose ? 1 : 0;
    A.read_name_off = reads->off.as<uint64_t>(); A.read_names = reads->blob.as<char>();
    A.ctg_name_off = contigs->off.as<uint64_t>(); A.ctg_names = contigs->blob.as<char>();
    A.read_len = reads->len.as<uint32_t>(); A.ctg_len = contigs->len.as<uint32_t>();
    A.tok_len = tok.as<uint32_t>(); A.hdr_len = hdr.as<uint32_t>(); A.paf_len = pl.as<uint32_t>();
    A.verbose = nullptr; A.paf = nullptr; A.ends = t->ends.as<HitRec>(); A.maps_out = t->maps.as<MapRec>();
    const uint64_t most = std::max<uint64_t>(std::max<uint64_t>(nh, nm), np);
    uint32_t tot[3] = {0, 0, 0};
    if (most) {
        ProfSpan sp(c, "format");
        const unsigned grid = (unsigned)((most + 255) / 256);
        hipLaunchKernelGGL(fmt_len_kernel, dim3(grid), dim3(256), 0, c->stream, A);
        if (!want_verbose) HIPCHK(c, hipMemsetAsync(A.hdr_len, 0, ((uint64_t)nm + 1) * 4, c->stream));
        if ((rc = device_scan(c, A.tok_len, A.tok_len, nh, nullptr)) || (rc = device_scan(c, A.hdr_len, A.hdr_len, nm, nullptr)) ||
            (rc = device_scan(c, A.paf_len, A.paf_len, np, nullptr)))
            return rc;
        HIPCHK(c, hipMemcpyAsync(&tot[0], A.tok_len + nh, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&tot[1], A.hdr_len + nm, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&tot[2], A.paf_len + np, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, main_wait(c));
        t->verbose_bytes = (uint64_t)tot[0] + tot[1];
        t->paf_bytes = tot[2];
        if (t->verbose_bytes >= 0xFFFFFFFFull) return fail(c, NTL_ERANGE, "ntl_mapres_format: more than 4 GB of text in one batch");
        if ((rc = t->verbose.alloc(c, t->verbose_bytes + 16)) || (rc = t->paf.alloc(c, t->paf_bytes + 16))) return rc;
        A.verbose = t->verbose.as<char>(); A.paf = t->paf.as<char>();
        hipLaunchKernelGGL(fmt_fill_kernel, dim3(grid), dim3(256), 0, c->stream, A);
        HIPCHK(c, hipGetLastError());
    }
    *out = t.release();
    return NTL_OK;
}

extern "C" void ntl_text_sizes(const ntl_text *t, uint64_t *verbose_bytes, uint64_t *paf_bytes, uint64_t *n_maps)
{
    if (verbose_bytes) *verbose_bytes = t ? t->verbose_bytes : 0;
    if (paf_bytes) *paf_bytes = t ? t->paf_bytes : 0;
    if (n_maps) *n_maps = t ? t->n_maps : 0;
}

extern "C" int ntl_text_download(const ntl_text *t, char *verbose, char *paf, ntl_mapping *maps, ntl_hit *ends)
{
    if (!t) return NTL_EINVAL;
    ntl_ctx *c = t->c;
    (void)hipSetDevice(c->device);
    if (verbose && t->verbose_bytes) HIPCHK(c, hipMemcpyAsync(verbose, t->verbose.p, t->verbose_bytes, hipMemcpyDeviceToHost, c->stream));
    if (paf && t->paf_bytes) HIPCHK(c, hipMemcpyAsync(paf, t->paf.p, t->paf_bytes, hipMemcpyDeviceToHost, c->stream));
    if (maps && t->n_maps) HIPCHK(c, hipMemcpyAsync(maps, t->maps.p, t->n_maps * sizeof(MapRec), hipMemcpyDeviceToHost, c->stream));
    if (ends && t->n_maps) HIPCHK(c, hipMemcpyAsync(ends, t->ends.p, t->n_maps * 2 * sizeof(HitRec), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, main_wait(c));
    return NTL_OK;
}

extern "C" void ntl_text_destroy(ntl_text *t)
{
    if (!t) return;
    (void)hipSetDevice(t->c->device);
    delete t;
}
