// api.hip -- the C-ABI of include/lpvspectral.h on top of the gfx950 kernels.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <array>
#include <thread>
#include <string>
#include <vector>

#include "lpvs_internal.h"

namespace lpvs {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- options -----------------------------------------------------------------------------------------------------------
static thread_local int g_opt[kOptCount] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static int option_from_env(int option) {
    auto is = [](const char *e, const char *v) { return e != nullptr && std::strcmp(e, v) == 0; };
    switch (option) {
    case LPVS_OPT_M_STORAGE: { const char *e = getenv("LPVS_M_STORAGE");
        return is(e, "mixed") ? LPVS_STORAGE_MIXED : is(e, "split") ? LPVS_STORAGE_SPLIT : is(e, "f64") ? LPVS_STORAGE_F64 : is(e, "mixed32") ? LPVS_STORAGE_MIXED32 : 0; }
    case LPVS_OPT_ITERATION: { const char *e = getenv("LPVS_ITERATION"); return is(e, "two") ? LPVS_ITERATION_TWO : is(e, "one") ? LPVS_ITERATION_ONE : 0; }
    case LPVS_OPT_GRAM_FORM: { const char *e = getenv("LPVS_GRAM_FORM"); return is(e, "ap") ? LPVS_GRAM_AP : (is(e, "krs") || is(e, "panel")) ? LPVS_GRAM_KRS : is(e, "kr") ? LPVS_GRAM_KR : 0; }   // (panel: the Fourier problems' dense form)
    case LPVS_OPT_NT_LOADS: { const char *e = getenv("LPVS_NT_LOADS"); return e == nullptr ? 0 : (e[0] == '1' ? LPVS_NT_ON : LPVS_NT_OFF); }
    case LPVS_OPT_SLOT_SUMS: { const char *e = getenv("LPVS_NUDFT"); return is(e, "direct") ? LPVS_SLOTS_DIRECT : is(e, "nufft") ? LPVS_SLOTS_NUFFT : 0; }
    case LPVS_OPT_WINDOW_CHUNK_MB: { const char *e = getenv("LPVS_WINDOW_CHUNK_MB"); if (!e) return 0; const double v = atof(e); return v <= 0 ? LPVS_WINDOW_UNCUT : (int)(v < 1 ? 1 : v); }
    case LPVS_OPT_WINDOWS_IN_FLIGHT: { const char *e = getenv("LPVS_WINDOWS_IN_FLIGHT"); if (!e) return 0; const int v = atoi(e); return v < 1 ? 1 : (v > 4 ? 4 : v); }
    case LPVS_OPT_RESERVE_CUS: { const char *e = getenv("LPVS_RESERVE_CUS"); if (!e) return 0; const int v = atoi(e); return v <= 0 ? LPVS_RESERVE_NONE : v; }
    case LPVS_OPT_XUPDATE_CORRECTION: { const char *e = getenv("LPVS_XUPDATE_CORRECTION"); if (!e) return 0; return (e[0] == '0' && e[1] == 0) ? LPVS_XCORR_OFF : LPVS_XCORR_ON; }   // ("0": off; any schedule: on)
    }
    return 0;
}
// Last-level (Infinity / MALL) cache of the device, in BYTES: the largest level-3 cache the KFD topology lists for a GPU node (a node
// whose properties say simd_count > 0 -- CPU nodes list their L3 the same way and must not set the window chunk plan; every GPU of a
// node is the same part).  256 MiB -- MI355X -- when the topology cannot be read.
double infinity_cache_bytes() {
    static const double bytes = [] {
        double best = 0;
        for (int node = 0; node < 64; ++node) {
            char npath[160];
            snprintf(npath, sizeof(npath), "/sys/class/kfd/kfd/topology/nodes/%d/properties", node);
            FILE *np_ = fopen(npath, "r");
            if (!np_) continue;
            char nkey[64]; long long nval = 0, simd = 0;
            while (fscanf(np_, "%63s %lld", nkey, &nval) == 2) if (!strcmp(nkey, "simd_count")) simd = nval;
            fclose(np_);
            if (simd <= 0) continue;                              // a CPU node
            for (int c = 0; c < 512; ++c) {
                char path[160];
                snprintf(path, sizeof(path), "/sys/class/kfd/kfd/topology/nodes/%d/caches/%d/properties", node, c);
                FILE *fp = fopen(path, "r");
                if (!fp) break;
                char key[64]; long long val = 0, level = 0, size_kb = 0;
                while (fscanf(fp, "%63s %lld", key, &val) == 2) {
                    if (!strcmp(key, "level")) level = val;
                    else if (!strcmp(key, "size")) size_kb = val;
                }
                fclose(fp);
                if (level == 3 && (double)size_kb * 1024.0 > best) best = (double)size_kb * 1024.0;
            }
        }
        return best >= (double)(16 << 20) ? best : 256.0 * 1048576.0;
    }();
    return bytes;
}
int option_in_effect(int option, int explicit_value) {
    if (option <= 0 || option >= kOptCount) return 0;
    if (explicit_value != 0) return explicit_value;
    if (g_opt[option] != 0) return g_opt[option];
    return option_from_env(option);          // (read per call: tests and tools switch it between handles)
}
void capture_default_options(int *opt) { for (int i = 0; i < kOptCount; ++i) opt[i] = g_opt[i]; }
static bool option_value_ok(int option, int value) {
    if (value == 0) return true;
    switch (option) {
    case LPVS_OPT_M_STORAGE: return value >= LPVS_STORAGE_MIXED && value <= LPVS_STORAGE_MIXED32;
    case LPVS_OPT_ITERATION: return value == LPVS_ITERATION_ONE || value == LPVS_ITERATION_TWO;
    case LPVS_OPT_GRAM_FORM: return value >= LPVS_GRAM_AP && value <= LPVS_GRAM_KR;
    case LPVS_OPT_NT_LOADS: return value == LPVS_NT_OFF || value == LPVS_NT_ON;
    case LPVS_OPT_SLOT_SUMS: return value == LPVS_SLOTS_NUFFT || value == LPVS_SLOTS_DIRECT;
    case LPVS_OPT_WINDOW_CHUNK_MB: return value == LPVS_WINDOW_UNCUT || (value >= 1 && value <= (1 << 20));
    case LPVS_OPT_WINDOWS_IN_FLIGHT: return value >= 1 && value <= 4;
    case LPVS_OPT_RESERVE_CUS: return value == LPVS_RESERVE_NONE || (value >= 1 && value <= 128);
    case LPVS_OPT_XUPDATE_CORRECTION: return value == LPVS_XCORR_ON || value == LPVS_XCORR_OFF;
    }
    return false;
}

// ---- caching device allocator -----------------------------------------------------------------------------
// Work buffers of a solve are tens of GiB (trig table, slabs, panels); hipMalloc/hipFree of that size costs
// milliseconds to seconds and synchronises the device.  Freed blocks are kept per device and handed out
// again (best fit within 25 % slack); the cache is capped (LPVS_POOL_GIB, default 128) and can be emptied
// with lpvs_release_cached_memory().  Every user synchronises its stream before releasing a buffer, so a
// cached block is idle by construction.
namespace {
struct Pool {
    std::mutex m;
    std::multimap<size_t, void *> free_[16];
    size_t cached[16] = {0};
    size_t cap() {
        static size_t c = [] { const char *e = getenv("LPVS_POOL_GIB"); const long g = e ? atol(e) : 128; return (size_t)(g < 0 ? 0 : g) << 30; }();
        return c;
    }
    void give(int dev, size_t n, void *p) {
        std::lock_guard<std::mutex> lk(m);
        if (n > cap()) { (void)hipFree(p); return; }
        free_[dev].emplace(n, p);
        cached[dev] += n;
        while (cached[dev] > cap() && !free_[dev].empty()) {   // evict the largest blocks first
            auto it = std::prev(free_[dev].end());
            cached[dev] -= it->first;
            (void)hipFree(it->second);
            free_[dev].erase(it);
        }
    }
    void flush() {
        std::lock_guard<std::mutex> lk(m);
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (int d = 0; d < 16; ++d) {
            if (free_[d].empty()) continue;
            (void)hipSetDevice(d);
            for (auto &kv : free_[d]) (void)hipFree(kv.second);
            free_[d].clear();
            cached[d] = 0;
        }
        (void)hipSetDevice(cur);
    }
};
Pool &pool() { static Pool p; return p; }
}  // namespace

size_t pool_cached_bytes(int dev) {
    if (dev < 0 || dev >= 16) return 0;
    std::lock_guard<std::mutex> lk(pool().m);
    return pool().cached[dev];
}

int32_t DevBuf::alloc(size_t nbytes) {
    release();
    if (nbytes == 0) nbytes = 16;
    nbytes = (nbytes + 255) & ~(size_t)255;
    int d = 0;
    (void)hipGetDevice(&d);
    dev = d;
    // the cache keys blocks by their true size, so a reused block may be up to 25 % larger than asked for
    if (d >= 0 && d < 16) {
        std::lock_guard<std::mutex> lk(pool().m);
        auto &fl = pool().free_[d];
        auto it = fl.lower_bound(nbytes);
        if (it != fl.end() && it->first <= nbytes + nbytes / 4 + (1 << 20)) {
            p = it->second; bytes = it->first;
            pool().cached[d] -= it->first;
            fl.erase(it);
            return LPVS_OK;
        }
    }
    hipError_t e = hipMalloc(&p, nbytes);
    if (e == hipErrorOutOfMemory) {   // give the cache back to the driver and retry once
        (void)hipGetLastError();
        pool().flush();
        e = hipMalloc(&p, nbytes);
    }
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipMalloc(%zu bytes) failed: %s", nbytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? LPVS_ENOMEM : LPVS_EDEVICE;
    }
    bytes = nbytes;
    return LPVS_OK;
}
void DevBuf::release() {
    if (p) {
        if (dev >= 0 && dev < 16) pool().give(dev, bytes, p);
        else (void)hipFree(p);
    }
    p = nullptr;
    bytes = 0;
}

bool is_device_ptr(const void *p) {
    if (p == nullptr) return false;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }  // plain (unregistered) host memory
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

int device_of_ptr(const void *p) {   // owning device of a device / managed allocation, -1 for host memory
    if (p == nullptr) return -1;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) ? at.device : -1;
}

int32_t copy_to_device(void *dst_dev, const void *src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return LPVS_OK;
    LPVS_HIP(hipMemcpyAsync(dst_dev, src, bytes, is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    LPVS_HIP(hipStreamSynchronize(s));  // the caller's host buffer is only valid during the call
    return LPVS_OK;
}
int32_t copy_from_device(void *dst, const void *src_dev, size_t bytes, hipStream_t s) {
    if (bytes == 0) return LPVS_OK;
    LPVS_HIP(hipMemcpyAsync(dst, src_dev, bytes, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    LPVS_HIP(hipStreamSynchronize(s));
    return LPVS_OK;
}

// A read-only argument made device-resident: aliases the caller's pointer when it already lives on the CURRENT device (the one the
// handle computes on); memory of another device is staged like host memory (a peer copy) -- kernels never dereference it.
struct DevArg {
    DevBuf own;
    const double *p = nullptr;
    int32_t set(const double *src, int64_t count, hipStream_t s) {
        const int owner = device_of_ptr(src);
        int cur = -1;
        if (owner >= 0) LPVS_HIP(hipGetDevice(&cur));
        if (owner >= 0 && owner == cur) { p = src; return LPVS_OK; }
        LPVS_TRY(own.alloc(sizeof(double) * (size_t)count));
        if (owner >= 0) {
            LPVS_HIP(hipMemcpyPeerAsync(own.p, cur, src, owner, sizeof(double) * (size_t)count, s));
            LPVS_HIP(hipStreamSynchronize(s));
        } else
        LPVS_TRY(copy_to_device(own.p, src, sizeof(double) * (size_t)count, s));
        p = own.as<double>();
        return LPVS_OK;
    }
};
// An output argument: device staging buffer when the caller's pointer is host memory.
struct DevOut {
    DevBuf own;
    double *p = nullptr;
    double *user = nullptr;
    size_t bytes = 0;
    int32_t set(double *dst, int64_t count) {
        user = dst; bytes = sizeof(double) * (size_t)count;
        if (is_device_ptr(dst)) { p = dst; return LPVS_OK; }
        LPVS_TRY(own.alloc(bytes));
        p = own.as<double>();
        return LPVS_OK;
    }
    int32_t finish(hipStream_t s) {
        if (p != user) return copy_from_device(user, p, bytes, s);
        LPVS_HIP(hipStreamSynchronize(s));
        return LPVS_OK;
    }
};

static int64_t check_freq_host(const double *f, int64_t Nf) {  // src/lsfft.jl:20-24
    for (int64_t i = 0; i < Nf; ++i)
        if (f[i] == 0.0) return i == 0 ? 1 : -1;
    return 0;
}
static int32_t fetch_host(std::vector<double> &h, const double *src, int64_t count) {
    h.resize((size_t)count);
    if (count == 0) return LPVS_OK;
    if (is_device_ptr(src)) { LPVS_HIP(hipMemcpy(h.data(), src, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost)); }
    else memcpy(h.data(), src, sizeof(double) * (size_t)count);
    return LPVS_OK;
}

// basis centres (src/utilities.jl:24-32); twice-precision range reproduced with long double
static void basis_centers(double lo, double hi, double amax, int64_t Nv, int coulomb, std::vector<double> &vc, double *gamma) {
    if (!coulomb) {
        vc.resize((size_t)Nv);
        for (int64_t j = 0; j < Nv; ++j)
            vc[j] = Nv > 1 ? (double)((long double)lo + (long double)j * ((long double)hi - (long double)lo) / (long double)(Nv - 1)) : lo;
        *gamma = (double)Nv / std::fabs(vc[0] - vc[Nv - 1]);
    } else {
        vc.resize((size_t)(2 * Nv));
        for (int64_t j = 0; j < Nv; ++j) {
            const double c = (double)((long double)(j + 1) * (long double)amax / (long double)(Nv + 1));
            vc[Nv + j] = c;
            vc[Nv - 1 - j] = -c;
        }
        *gamma = (double)(2 * Nv) / std::fabs(vc[0] - vc[2 * Nv - 1]);
    }
}

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    int32_t init() { LPVS_HIP(hipEventCreate(&a)); LPVS_HIP(hipEventCreate(&b)); return LPVS_OK; }
    void destroy() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); a = b = nullptr; }
    double ms() const { float t = 0; if (hipEventElapsedTime(&t, a, b) != hipSuccess) { (void)hipGetLastError(); return 0; } return t; }
};

// Declared AFTER the temporaries of a scope (so destroyed BEFORE them): on an early error return the stream is drained
// before the temporaries' blocks go back to the process-wide cache, where another handle's stream could pick them up.
}  // namespace lpvs

using namespace lpvs;

// Streams and events are not free to create and destroy (about 1 ms per handle): handles borrow them from a small
// per-process cache and give them back idle.
struct StreamBundle {
    int device = -1;
    hipStream_t stream = nullptr;
    EventPair ev[4];
    SweepAux aux;
    ~StreamBundle() {
        for (auto &e : ev) e.destroy();
        if (stream) (void)hipStreamDestroy(stream);
    }
};
static std::mutex g_bundle_mu;
static std::vector<StreamBundle *> g_bundles;
static StreamBundle *bundle_acquire(int device) {
    {
        std::lock_guard<std::mutex> g(g_bundle_mu);
        for (size_t i = 0; i < g_bundles.size(); ++i)
            if (g_bundles[i]->device == device) { StreamBundle *b = g_bundles[i]; g_bundles.erase(g_bundles.begin() + (long)i); return b; }
    }
    // Idle bundles are destroyed by an exit handler registered HERE, i.e. after the HIP runtime's own: it runs before the runtime is
    // torn down (a CU-masked stream still alive at that point crashed the process at exit under rocprofv3).
    static std::once_flag exit_hook;
    std::call_once(exit_hook, [] {
        (void)atexit([] {
            std::lock_guard<std::mutex> g(g_bundle_mu);
            for (StreamBundle *q : g_bundles) { (void)hipSetDevice(q->device); delete q; }
            g_bundles.clear();
        });
    });
    StreamBundle *b = new (std::nothrow) StreamBundle();
    if (!b) return nullptr;
    b->device = device;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); delete b; return nullptr; }
    for (auto &e : b->ev) if (e.init() != LPVS_OK) { delete b; return nullptr; }
    return b;
}
static void bundle_release(StreamBundle *b) {
    if (!b) return;
    (void)hipStreamSynchronize(b->stream);
    if (b->aux.side) (void)hipStreamSynchronize(b->aux.side);
    std::lock_guard<std::mutex> g(g_bundle_mu);
    if (g_bundles.size() < 8) g_bundles.push_back(b); else delete b;
}

struct lpvs_problem {
    int kind = 0;  // 0 fourier, 1 lpv, 2 explicit gram
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n = 0, np = 0, N = 0, Nf = 0, nb = 0, zerofreq = 0;
    int64_t ns = 1;   // signals sharing the regressor (right-hand sides); state vectors are [ns][np]
    DevBuf G, b, M, x, z, u, rhs, bs, scratch, status, work, istat, part, Mp;
    bool M_valid = false; double M_shift = 0;
    bool Mp_valid = false;   // Mp is the packed copy of the CURRENT M (cleared whenever M is recomputed or G changes)
    int64_t Mp_fixed_diag = 0;   // ... of which on the diagonal (type 2)
    DevBuf sm_ctl;               // control words of the one-launch iteration of small problems (AdmmParams::sm_ctl)
    int Mp_mode = 0;         // storage of Mp: kMpF64 / kMpF32 / kMpSplit / kMpMixed (see make_params)
    double Mp_stream_bytes = 0;   // bytes of Mp one mat-vec launch reads (mixed storage: depends on the tile formats chosen)
    int64_t Mp_fixed_tiles = 0;
    bool Mp_demoted = false;      // mixed storage was asked for, but fewer than half of the tiles qualified: stored as split
    DevBuf xb; bool offset_form = false;   // xb = M * (signed b), computed at admm_init from the full-precision M (AdmmParams::xb)
    // the x-update's systematic error removed (admm.hip, launch_xupdate_correction): xb0 = the refined offset vector, xb = xb0 - E (x_k - xb0)
    // re-formed after the iterations k = 1, 2, 4, 8, ... (k_enq = iterations enqueued since lpvs_admm_init / lpvs_admm_set_state)
    DevBuf xb_corr, nib_rhs, nib_part, nib_acc; int nib_period = 0, nib_ramp = 0; double n_nib = 0, nib_us = 0;   // refreshes enqueued by the timed runs; one refresh stand-alone (lpvs_admm_time_matvec)   // 32-bit reads with a stale nibble product (AdmmParams::nib_period): the offset vector without its nibble term, scratch
    DevBuf xb0, corr; int xcorr_base = 0, xcorr_every = 0; long long k_enq = 0; bool xb_refined = false, xcorr_double = false, xcorr_early = false;
    // the schedule: after the iterations base^j (xcorr_base >= 2), or after iteration 16 and every xcorr_every-th one; 0 0 = no correction
    long long next_correction(long long k) const {
        if (xcorr_every > 0 && xcorr_double) {   // 16 (xcorr_early: 1, 2, 4, 8, 16), then xcorr_every and its doublings: the iterates move ever more slowly
            if (xcorr_early && k < 16) { long long q = 1; while (q <= k) q *= 2; return q; }
            if (k < 16 && xcorr_every > 16) return 16;
            long long q = xcorr_every;
            while (q <= k) q *= 2;
            return q;
        }
        if (xcorr_every > 0) { if (k < 16 && xcorr_every > 16) return 16; return (k / xcorr_every + 1) * (long long)xcorr_every; }
        long long q = 1;
        while (q <= k) q *= xcorr_base;
        return q;
    }
    bool xcorr() const { return xcorr_base >= 2 || xcorr_every > 0; }
    bool Mp_read32 = false; // the iteration reads the 32 leading bits of the fixed-point tiles only (storage mixed32 on a corrected handle)
    int Mp_fix_bits = 36;   // significant bits of the fixed-point tiles of the packed copy (32: nibbles zero and not read)
    double Mp_rowsum = 0;   // largest absolute row sum of M over its valid rows, left by the mixed packing pass (0: not known)
    DevBuf fi; long long fi_sync = -1; double fi_R = 0, fi_xbmax = 0;      // one-launch iteration (AdmmParams::fi): its records are those of iteration fi_sync
    int prox_kind = LPVS_PROX_L1; double prox_param = 1.0; int64_t group_len = 0;
    double mu = 0.05, tol = 1e-5; int sign = 1; bool inited = false;
    // timing (ms) -- see lpvs_problem_get_timing
    double t_basis = 0, t_gram = 0, t_reduce = 0, t_factor = 0, t_admm = 0, gram_launches = 0, gram_flops = 0, admm_iters_timed = 0, gram_form = 0;
    double t_xcorr = 0, n_xcorr = 0;       // the x-update corrections inside t_admm: their time (HIP events around each) and count
    std::vector<hipEvent_t> xc_ev;         // event pairs of the corrections of the running lpvs_admm_run call (read after its final synchronisation, then reused)
    EventPair ev[4];          // copies of the bundle's events (owned by `res`)
    StreamBundle *res = nullptr;   // stream, events and the factorisation's side stream, borrowed from the cache
    bool f32 = false;     // created through an _f32 entry point: the ADMM mat-vec streams a single-precision copy of M
    int opt[kOptCount] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // LPVS_OPT_*: explicit values of this handle (the creating thread's defaults at creation, then set_option)
    // launch-bound regime (small n): a chunk of ADMM iterations captured once into a hipGraph and replayed
    hipGraphExec_t admm_graph = nullptr;
    int64_t admm_graph_iters = 0;
    void drop_graph() { if (admm_graph) (void)hipGraphExecDestroy(admm_graph); admm_graph = nullptr; admm_graph_iters = 0; }
    ~lpvs_problem() {
        for (hipEvent_t e : xc_ev) (void)hipEventDestroy(e);
        drop_graph();
        bundle_release(res);
    }
};

namespace {

int32_t problem_begin(int32_t device, lpvs_problem **out, lpvs_problem **hp) {
    if (out == nullptr) { set_error("out handle pointer is NULL"); return LPVS_EARGUMENT; }
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible (the gfx950 path has no CPU fallback)");
        return LPVS_EDEVICE;
    }
    if (device < 0 || device >= count) { set_error("device %d out of range [0,%d)", device, count); return LPVS_EDEVICE; }
    LPVS_HIP(hipSetDevice(device));
    lpvs_problem *h = new (std::nothrow) lpvs_problem();
    if (!h) return LPVS_ENOMEM;
    h->device = device;
    h->res = bundle_acquire(device);
    if (!h->res) { delete h; set_error("stream / event creation failed"); return LPVS_EDEVICE; }
    h->stream = h->res->stream;
    for (int i = 0; i < 4; ++i) h->ev[i] = h->res->ev[i];
    capture_default_options(h->opt);
    *hp = h;
    return LPVS_OK;
}

int32_t alloc_state(lpvs_problem *h) {
    const size_t v = sizeof(double) * (size_t)h->np * (size_t)h->ns;
    LPVS_TRY(h->x.alloc(v)); LPVS_TRY(h->z.alloc(v)); LPVS_TRY(h->u.alloc(v)); LPVS_TRY(h->rhs.alloc(v));
    LPVS_TRY(h->bs.alloc(v)); LPVS_TRY(h->scratch.alloc(2 * v));
    LPVS_TRY(h->status.alloc(sizeof(AdmmStatus) * (size_t)h->ns)); LPVS_TRY(h->istat.alloc(sizeof(int)));
    if (h->np < kSymmetricMinNp) { LPVS_TRY(h->sm_ctl.alloc(sizeof(int) * 4 * (size_t)h->ns)); LPVS_HIP(hipMemsetAsync(h->sm_ctl.p, 0, h->sm_ctl.bytes, h->stream)); }
    LPVS_TRY(h->part.alloc(sizeof(double) * symv_part_doubles(h->np, h->ns)));
    LPVS_HIP(hipMemsetAsync(h->x.p, 0, v, h->stream));
    LPVS_HIP(hipMemsetAsync(h->status.p, 0, sizeof(AdmmStatus) * (size_t)h->ns, h->stream));
    return LPVS_OK;
}

// device [ns][np] (first n of every row valid)  <->  caller's n x ns column-major array
int32_t copy_state_out(lpvs_problem *h, const void *dev, double *dst) {
    LPVS_HIP(hipMemcpy2DAsync(dst, sizeof(double) * (size_t)h->n, dev, sizeof(double) * (size_t)h->np, sizeof(double) * (size_t)h->n,
                              (size_t)h->ns, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    LPVS_HIP(hipStreamSynchronize(h->stream));
    return LPVS_OK;
}
int32_t copy_state_in(lpvs_problem *h, void *dev, const double *src) {
    LPVS_HIP(hipMemcpy2DAsync(dev, sizeof(double) * (size_t)h->np, src, sizeof(double) * (size_t)h->n, sizeof(double) * (size_t)h->n,
                              (size_t)h->ns, is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    LPVS_HIP(hipStreamSynchronize(h->stream));
    return LPVS_OK;
}

// storage of the tile-packed inverse streamed by the ADMM mat-vec of large problems
enum { kMpNone = 0, kMpF64 = 1, kMpF32 = 2, kMpSplit = 3, kMpMixed = 4 };
// LPVS_M_STORAGE = mixed (default) | split | f64: how a double-precision handle stores the tile-packed inverse its mat-vec streams.
// split = float head + 16-bit tail (6 bytes, 40 significant bits, admm.hip); mixed = the same, except that
// tiles whose entries are all small against max|M| (the off-diagonal tiles of the LPV / Fourier inverses) are 36-bit fixed point
// (4.53 bytes per element); _f32 handles always stream floats.  Handles with several right-hand sides decode either format on
// the way into the LDS image their matrix-core tile product reads (their diagonal tiles always stay 6-byte).
int mp_mode_for(const lpvs_problem *h) {
    if (h->np < kSymmetricMinNp) return kMpNone;
    if (h->f32) return kMpF32;
    const int st = option_in_effect(LPVS_OPT_M_STORAGE, h->opt[LPVS_OPT_M_STORAGE]);
    if (st == LPVS_STORAGE_F64) return kMpF64;
    if (h->ns > 1 && h->np > 49152) return kMpF64;   // the streaming multi-signal kernel addresses its partials with 31-bit byte offsets
    if (st == LPVS_STORAGE_SPLIT) return kMpSplit;
    if (h->ns > 1 && !multi_signal_fixed_tiles_ok(h->np)) return kMpSplit;
    return kMpMixed;   // (several right-hand sides: diagonal tiles always float-head; lpvs_admm_init)
}

AdmmParams make_params(const lpvs_problem *h) {
    const bool sym = h->np >= kSymmetricMinNp;
    AdmmParams p{h->M.as<double>(), h->np, h->n, h->bs.as<double>(), h->x.as<double>(), h->z.as<double>(), h->u.as<double>(),
                 h->rhs.as<double>(), h->mu, h->tol, h->prox_kind, h->prox_param, h->group_len, h->status.as<AdmmStatus>(),
                 h->scratch.as<double>(), h->part.as<double>(), sym ? h->Mp.as<double>() : nullptr, (int)h->ns};
    p.mp_f32 = sym && h->Mp_mode == kMpF32 ? 1 : 0;
    p.mp_split = sym && (h->Mp_mode == kMpSplit || h->Mp_mode == kMpMixed) ? 1 : 0;
    p.mp_types = sym && h->Mp_mode == kMpMixed ? h->Mp.as<unsigned char>() + 6 * symv_packed_doubles(h->np) : nullptr;
    p.mp_fix32 = p.mp_types != nullptr && (h->Mp_fix_bits <= 32 || h->nib_period > 0) ? 1 : 0;
    p.nib_period = p.mp_types != nullptr ? h->nib_period : 0; p.nib_ramp = h->nib_ramp; p.xb_corr = h->xb_corr.as<double>(); p.nib_rhs = h->nib_rhs.as<double>(); p.nib_part = h->nib_part.as<double>(); p.nib_acc = h->nib_acc.as<long long>();
    p.xb = sym && h->offset_form ? h->xb.as<double>() : nullptr;
    p.fi = sym && h->offset_form && h->ns == 1 && (h->Mp_mode == kMpMixed || h->Mp_mode == kMpF32) && h->fi.p ? h->fi.as<double>() : nullptr;
    p.fi_R = h->fi_R; p.fi_xbmax = h->fi_xbmax;
    p.fi_prefetch_all = sym && h->Mp_mode == kMpMixed && h->Mp_fixed_tiles == (int64_t)(symv_packed_doubles(h->np) / (128 * 128)) ? 1 : 0;
    p.opt_iteration = h->opt[LPVS_OPT_ITERATION]; p.opt_nt_loads = h->opt[LPVS_OPT_NT_LOADS];
    p.sm_ctl = !sym && h->sm_ctl.p ? h->sm_ctl.as<int>() : nullptr;
    return p;
}

// M = (G + shift I)^-1 on the padded np x np system (pad diagonal = 1), cached by shift.
int32_t factorize(lpvs_problem *h, double shift) {
    if (h->M_valid && h->M_shift == shift) return LPVS_OK;
    const int64_t np = h->np, n = h->n;
    h->Mp_valid = false; h->Mp_rowsum = 0;
    if (!h->M.p) LPVS_TRY(h->M.alloc(sizeof(double) * (size_t)np * (size_t)np));
    if (!h->work.p) LPVS_TRY(h->work.alloc(spd_inverse_work_bytes(np)));
    LPVS_HIP(hipEventRecord(h->ev[3].a, h->stream));
    LPVS_HIP(hipMemcpyAsync(h->M.p, h->G.p, sizeof(double) * (size_t)np * (size_t)np, hipMemcpyDeviceToDevice, h->stream));
    LPVS_TRY(launch_add_diag(h->M.as<double>(), np, n, shift, h->stream));
    LPVS_TRY(spd_inverse_inplace(h->M.as<double>(), np, h->work.as<double>(), h->istat.as<int>(), h->stream, &h->res->aux));
    LPVS_HIP(hipEventRecord(h->ev[3].b, h->stream));
    int st = 0;
    LPVS_HIP(hipMemcpyAsync(&st, h->istat.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    LPVS_HIP(hipStreamSynchronize(h->stream));
    h->t_factor += h->ev[3].ms();
    if (st != 0) {
        h->M_valid = false;
        set_error("(G + %.3g I) is not positive definite (non-positive pivot in the block sweep)", shift);
        return LPVS_ENUMERIC;
    }
    h->M_valid = true; h->M_shift = shift;
    return LPVS_OK;
}

}  // namespace

// Common tail of the panel-form constructors: P is the k-major regressor panel [Npad][ld].
// fill(P) must enqueue the kernel(s) that write rows [0,N) of the panel on the handle's stream.
template <class Fill>
static int32_t create_panel_problem(lpvs_problem *h, const double *y, const double *W, int64_t N, Fill fill) {
    hipStream_t s = h->stream;
    h->np = round_up(h->n, 128);
    const GramPlan pl = make_gram_plan(h->n, N);
    const int64_t Npad = pl.ksplit * pl.rows_per_chunk;
    const int64_t ld = round_up(h->n, 256);
    DevArg dy;
    LPVS_TRY(dy.set(y, N, s));
    LPVS_TRY(h->G.alloc(sizeof(double) * (size_t)h->np * (size_t)h->np));
    LPVS_TRY(h->b.alloc(sizeof(double) * (size_t)h->np));
    LPVS_HIP(hipMemsetAsync(h->G.p, 0, h->G.bytes, s));
    LPVS_HIP(hipMemsetAsync(h->b.p, 0, h->b.bytes, s));
    LPVS_TRY(alloc_state(h));

    DevBuf P, Wp, slab, scr;
    DrainOnExit drain(s);
    LPVS_TRY(P.alloc(sizeof(double) * (size_t)Npad * (size_t)ld));
    if (Npad > N) LPVS_HIP(hipMemsetAsync(P.as<double>() + N * ld, 0, sizeof(double) * (size_t)(Npad - N) * (size_t)ld, s));
    const double *Wdev = nullptr;
    if (W != nullptr) {  // padded copy: pad rows carry weight 0
        LPVS_TRY(Wp.alloc(sizeof(double) * (size_t)Npad));
        LPVS_HIP(hipMemsetAsync(Wp.p, 0, Wp.bytes, s));
        LPVS_TRY(copy_to_device(Wp.p, W, sizeof(double) * (size_t)N, s));
        Wdev = Wp.as<double>();
    }
    LPVS_HIP(hipEventRecord(h->ev[0].a, s));
    LPVS_TRY(fill(P.as<double>(), ld));
    LPVS_HIP(hipEventRecord(h->ev[0].b, s));
    LPVS_TRY(slab.alloc(pl.slab_bytes));
    LPVS_TRY(scr.alloc(rhs_scratch_bytes(N, h->n)));
    LPVS_HIP(hipEventRecord(h->ev[1].a, s));
    LPVS_TRY(launch_gram_panel(pl, P.as<double>(), ld, Wdev, slab.as<double>(), s));
    LPVS_HIP(hipEventRecord(h->ev[1].b, s));
    LPVS_HIP(hipEventRecord(h->ev[2].a, s));
    LPVS_TRY(launch_gram_reduce(pl, slab.as<double>(), h->G.as<double>(), h->np, s));
    LPVS_TRY(launch_rhs_panel(P.as<double>(), ld, h->n, Wdev, dy.p, N, h->b.as<double>(), scr.as<double>(), scr.bytes, s));
    LPVS_HIP(hipEventRecord(h->ev[2].b, s));
    LPVS_HIP(hipStreamSynchronize(s));
    h->t_basis = h->ev[0].ms(); h->t_gram = h->ev[1].ms(); h->t_reduce = h->ev[2].ms();
    h->gram_launches = (double)pl.tiles * 128.0 * 256.0 * 2.0 * (double)(pl.ksplit * pl.rows_per_chunk);   // flops the MFMA core issues
    h->gram_form = 3;
    h->gram_flops = (double)N * (double)h->n * (double)(h->n + 1);
    return LPVS_OK;
}

extern "C" {

int32_t lpvs_version(void) { return LPVS_VERSION; }

int32_t lpvs_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return c;
}

const char *lpvs_last_error(void) { return g_err; }

int32_t lpvs_release_cached_memory(void) {
    pool().flush();
    std::vector<StreamBundle *> idle;
    { std::lock_guard<std::mutex> g(g_bundle_mu); idle.swap(g_bundles); }
    for (StreamBundle *b : idle) delete b;           // idle streams / events of destroyed handles
    release_panel_plans();                           // the panel walk's cached device tables (admm.hip)
    return LPVS_OK;
}

int32_t lpvs_set_default_option(int32_t option, int32_t value) {
    if (option <= 0 || option >= kOptCount || !option_value_ok(option, value)) { set_error("unknown option %d or value %d", option, value); return LPVS_EARGUMENT; }
    g_opt[option] = value;
    return LPVS_OK;
}
int32_t lpvs_get_default_option(int32_t option, int32_t *value) {
    if (option <= 0 || option >= kOptCount || !value) { set_error("unknown option %d or NULL argument", option); return LPVS_EARGUMENT; }
    *value = g_opt[option];
    return LPVS_OK;
}
int32_t lpvs_problem_set_option(lpvs_problem *h, int32_t option, int32_t value) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    if (option <= 0 || option >= kOptCount || !option_value_ok(option, value)) { set_error("unknown option %d or value %d", option, value); return LPVS_EARGUMENT; }
    if (option == LPVS_OPT_GRAM_FORM || option == LPVS_OPT_SLOT_SUMS || (option >= LPVS_OPT_WINDOW_CHUNK_MB && option != LPVS_OPT_XUPDATE_CORRECTION)) {
        set_error("option %d is chosen when a handle is constructed (or belongs to calls without a handle): set it with lpvs_set_default_option", option);
        return LPVS_ESTATE;
    }
    if (h->opt[option] != value) {
        h->opt[option] = value;
        if (option == LPVS_OPT_M_STORAGE || option == LPVS_OPT_XUPDATE_CORRECTION) h->inited = false;   // the packed copy / the schedule are set up by the next lpvs_admm_init
        h->drop_graph();
    }
    return LPVS_OK;
}
int32_t lpvs_problem_get_option(lpvs_problem *h, int32_t option, int32_t *value) {
    if (!h || !value) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    if (option <= 0 || option >= kOptCount) { set_error("unknown option %d", option); return LPVS_EARGUMENT; }
    *value = option_in_effect(option, h->opt[option]);
    return LPVS_OK;
}

int32_t lpvs_check_freq_f64(const double *f, int64_t Nf, int64_t *zerofreq) {
    std::vector<double> hf;
    LPVS_TRY(fetch_host(hf, f, Nf));
    const int64_t z = check_freq_host(hf.data(), Nf);
    if (z < 0) { set_error("If zero frequency is included it must be the first frequency"); return LPVS_EARGUMENT; }
    if (zerofreq) *zerofreq = z;
    return LPVS_OK;
}

int32_t lpvs_fourier_regressor_f64(const double *t, int64_t N, const double *f, int64_t Nf, double *A_out, int64_t *zerofreq) {
    int64_t z = 0;
    LPVS_TRY(lpvs_check_freq_f64(f, Nf, &z));
    if (zerofreq) *zerofreq = z;
    if (lpvs_device_count() == 0) { set_error("no HIP device visible (the gfx950 path has no CPU fallback)"); return LPVS_EDEVICE; }
    hipStream_t s = nullptr;
    const int64_t nreg = z ? 2 * Nf - 1 : 2 * Nf;
    DevArg dt, df; DevOut dA;
    LPVS_TRY(dt.set(t, N, s)); LPVS_TRY(df.set(f, Nf, s)); LPVS_TRY(dA.set(A_out, N * nreg));
    LPVS_TRY(launch_fourier_regressor_colmajor(dt.p, N, df.p, Nf, (int)z, dA.p, s));
    return dA.finish(s);
}

int32_t lpvs_basis_activation_f64(const double *V, int64_t N, int64_t Nv, int32_t normalize, int32_t coulomb, double *K_out) {
    if (lpvs_device_count() == 0) { set_error("no HIP device visible (the gfx950 path has no CPU fallback)"); return LPVS_EDEVICE; }
    if (N <= 0 || Nv <= 0) { set_error("N and Nv must be positive"); return LPVS_EARGUMENT; }
    hipStream_t s = nullptr;
    const int64_t nb = coulomb ? 2 * Nv : Nv;
    DevArg dV; LPVS_TRY(dV.set(V, N, s));
    double lo, hi, am, gamma; std::vector<double> vc;
    LPVS_TRY(device_minmax(dV.p, N, &lo, &hi, &am, s));
    basis_centers(lo, hi, am, Nv, coulomb, vc, &gamma);
    DevBuf dvc, Kt; LPVS_TRY(dvc.alloc(sizeof(double) * vc.size()));
    LPVS_TRY(copy_to_device(dvc.p, vc.data(), sizeof(double) * vc.size(), s));
    LPVS_TRY(Kt.alloc(sizeof(double) * (size_t)N * (size_t)nb));
    LPVS_TRY(launch_basis_table(dV.p, N, dvc.as<double>(), nb, gamma, normalize, coulomb, Kt.as<double>(), nb, s));
    // row-major [N][nb] -> column-major N x nb on the host side of the boundary (tiny table)
    std::vector<double> hk((size_t)N * nb), ho((size_t)N * nb);
    LPVS_TRY(copy_from_device(hk.data(), Kt.p, sizeof(double) * hk.size(), s));
    for (int64_t n = 0; n < N; ++n) for (int64_t j = 0; j < nb; ++j) ho[n + j * N] = hk[n * nb + j];
    if (is_device_ptr(K_out)) { LPVS_HIP(hipMemcpy(K_out, ho.data(), sizeof(double) * ho.size(), hipMemcpyHostToDevice)); }
    else memcpy(K_out, ho.data(), sizeof(double) * ho.size());
    return LPVS_OK;
}

// shared by lpvs_lpv_regressor_f64 and lpvs_problem_create_lpv_f64: T and K tables with Npad rows
static int32_t build_lpv_tables(const double *dX, const double *dV, int64_t N, int64_t Npad, const double *dw, int64_t Nf,
                                int64_t Nv, int normalize, int coulomb, DevBuf &T, DevBuf &K, int64_t *ldk_out, hipStream_t s,
                                const double *ranges = nullptr) {
    const int64_t nb = coulomb ? 2 * Nv : Nv, ldk = nb + 1;
    double lo, hi, am, gamma; std::vector<double> vc;
    if (ranges) { lo = ranges[0]; hi = ranges[1]; am = ranges[2]; }
    else LPVS_TRY(device_minmax(dV, N, &lo, &hi, &am, s));
    basis_centers(lo, hi, am, Nv, coulomb, vc, &gamma);
    DevBuf dvc; LPVS_TRY(dvc.alloc(sizeof(double) * vc.size()));
    LPVS_TRY(copy_to_device(dvc.p, vc.data(), sizeof(double) * vc.size(), s));
    LPVS_TRY(T.alloc(sizeof(double2) * (size_t)Npad * (size_t)Nf));
    LPVS_TRY(K.alloc(sizeof(double) * (size_t)Npad * (size_t)ldk));
    if (Npad > N) {  // zero pad rows: they contribute nothing to the contraction
        LPVS_HIP(hipMemsetAsync(T.as<double2>() + N * Nf, 0, sizeof(double2) * (size_t)(Npad - N) * (size_t)Nf, s));
        LPVS_HIP(hipMemsetAsync(K.as<double>() + N * ldk, 0, sizeof(double) * (size_t)(Npad - N) * (size_t)ldk, s));
    }
    LPVS_TRY(launch_trig_table(dX, N, dw, Nf, T.as<double2>(), s));
    LPVS_TRY(launch_basis_table(dV, N, dvc.as<double>(), nb, gamma, normalize, coulomb, K.as<double>(), ldk, s));
    LPVS_HIP(hipStreamSynchronize(s));  // dvc is released on return
    *ldk_out = ldk;
    return LPVS_OK;
}

int32_t lpvs_lpv_regressor_f64(const double *X, const double *V, int64_t N, const double *w, int64_t Nf, int64_t Nv,
                               int32_t normalize, int32_t coulomb, int32_t permuted, double *Phi_out) {
    if (lpvs_device_count() == 0) { set_error("no HIP device visible (the gfx950 path has no CPU fallback)"); return LPVS_EDEVICE; }
    if (N <= 0 || Nf <= 0 || Nv <= 0) { set_error("N, Nf and Nv must be positive"); return LPVS_EARGUMENT; }
    hipStream_t s = nullptr;
    const int64_t nb = coulomb ? 2 * Nv : Nv;
    DevArg dX, dV, dw; DevOut dP;
    LPVS_TRY(dX.set(X, N, s)); LPVS_TRY(dV.set(V, N, s)); LPVS_TRY(dw.set(w, Nf, s));
    LPVS_TRY(dP.set(Phi_out, N * 2 * Nf * nb));
    DevBuf T, K; int64_t ldk;
    LPVS_TRY(build_lpv_tables(dX.p, dV.p, N, N, dw.p, Nf, Nv, normalize, coulomb, T, K, &ldk, s));
    LPVS_TRY(launch_lpv_regressor_colmajor(T.as<double2>(), K.as<double>(), ldk, N, Nf, nb, permuted, dP.p, s));
    return dP.finish(s);
}

// Slot tables of the structured Gram (nudft.hip) for frequencies w_f = a + f*D + eps_f: exact progressions in
// double-double, each family padded to a multiple of 8:
//   [0, nf8): m*D (differences), [nf8, nf8+s8): 2a + s*D (sums); the right-hand side uses a + f*D, f < nf8.
// MERGED layout: when 2a is (up to rounding) an integer multiple j0 < Nf of D -- default_freqs (a = 0), the README grid
// w = D*(1..Nf) (a = D) -- the sum frequencies (j0 + s) D are multiples of D like the differences, so ONE progression
// j*D, j = 0 .. j0+2Nf-2 serves both families: 2Nf-1+j0 slots instead of 3Nf-1 (a third less work); the sum of (f, f') is
// slot s0 + f + f' with s0 = j0, and the rounding residual delta = 2a - j0*D joins the first-order correction.
struct ApSlots {
    bool ok = false;
    double emax = 0;
    int64_t nf8 = 0, s8 = 0, nsl = 0;
    bool merged = false;   // one progression j*D serves differences and sums (see above)
    int64_t s0 = 0;        // slot of the sum frequency 2a (split layout: nf8; merged: j0)
    double delta = 0;      // merged layout: 2a - j0*D
    std::vector<double> eps, om_hi, om_lo, omr_hi, omr_lo;
    ApStep step{};
};
// Single-precision callers (_f32 entry points): the frequency grid is a progression rounded to FLOATS, so its residuals are up
// to 2^-24 |w| and |eps| max|x| reaches 0.2 rad at the cfg3 size -- far beyond the double-precision admission bound.  But a
// Float32 run of the reference evaluates fl32(w x) itself, a phase error of up to 2^-24 |w x|: snapping w to the exact
// progression its floats were rounded from stays inside that error.  While an _f32 constructor runs, residuals up to
// 2^-23 max|w| (one float ulp) are admitted and dropped.
static thread_local bool g_f32_admission = false;

static ApSlots make_ap_slots(const std::vector<double> &hw, double xam) {
    ApSlots sl;
    const int64_t Nf = (int64_t)hw.size();
    const long double a0 = hw[0], D = Nf > 1 ? ((long double)hw[Nf - 1] - (long double)hw[0]) / (long double)(Nf - 1) : 0.0L;
    sl.eps.resize((size_t)Nf);
    for (int64_t f = 0; f < Nf; ++f) {
        sl.eps[f] = (double)((long double)hw[f] - (a0 + (long double)f * D));
        sl.emax = std::fmax(sl.emax, std::fabs(sl.eps[f]));
    }
    double wam = 0;
    for (double v : hw) wam = std::fmax(wam, std::fabs(v));
    sl.ok = std::isfinite(sl.emax) && std::isfinite(xam) && sl.emax * xam <= 1e-7;   // second-order term (eps x)^2/2 <= 5e-15
    if (!sl.ok && g_f32_admission && std::isfinite(sl.emax) && std::isfinite(xam) && sl.emax <= 0x1p-23 * wam) {
        // large residual phases: snapped grid, NO first-order term -- G is then exactly the Gram of the regressor at the snapped
        // frequencies (positive semidefinite by construction), which a first-order expansion with |eps x| ~ 0.2 would not guarantee
        sl.ok = true;
        if (sl.emax * xam > 3e-3) for (auto &e : sl.eps) e = 0.0;   // small residual phases keep the first-order term (error (eps x)^2/2 <= 5e-6)
    }
    if (!sl.ok) return sl;
    sl.nf8 = round_up(Nf, 8); sl.s8 = round_up(2 * Nf - 1, 8); sl.nsl = sl.nf8 + sl.s8;
    auto split = [](long double v, double &hi, double &lo) { hi = (double)v; lo = (double)(v - (long double)hi); };
    sl.omr_hi.resize((size_t)sl.nf8); sl.omr_lo.resize((size_t)sl.nf8);
    sl.s0 = sl.nf8;
    bool merged = false;
    if (Nf > 1 && D != 0.0L) {
        const long double j0r = 2.0L * a0 / D;
        const long long j0 = llroundl(j0r);
        const long double delta = 2.0L * a0 - (long double)j0 * D;
        if (j0 >= 0 && j0 < Nf && std::fabs((double)delta) * xam <= 1e-7) {
            merged = true; sl.merged = true;
            sl.s0 = j0; sl.delta = (double)delta;
            sl.nsl = round_up(j0 + 2 * Nf - 1, 8);
            sl.s8 = sl.nsl;
        }
    }
    sl.om_hi.resize((size_t)sl.nsl); sl.om_lo.resize((size_t)sl.nsl);
    if (merged) {
        for (int64_t j = 0; j < sl.nsl; ++j) split((long double)j * D, sl.om_hi[j], sl.om_lo[j]);
    } else {
        for (int64_t m = 0; m < sl.nf8; ++m) split((long double)m * D, sl.om_hi[m], sl.om_lo[m]);
        for (int64_t q = 0; q < sl.s8; ++q) split(2.0L * a0 + (long double)q * D, sl.om_hi[sl.nf8 + q], sl.om_lo[sl.nf8 + q]);
    }
    for (int64_t f = 0; f < sl.nf8; ++f) split(a0 + (long double)f * D, sl.omr_hi[f], sl.omr_lo[f]);
    for (int b = 0; b < 8; ++b) split((long double)b * D, sl.step.hi[b], sl.step.lo[b]);
    return sl;
}
// device copies of the slot tables
struct ApSlotsDev {
    DevBuf hi, lo, rhi, rlo, eps;
    int32_t upload(const ApSlots &sl, hipStream_t s) {
        LPVS_TRY(hi.alloc(sizeof(double) * (size_t)sl.nsl)); LPVS_TRY(lo.alloc(sizeof(double) * (size_t)sl.nsl));
        LPVS_TRY(rhi.alloc(sizeof(double) * (size_t)sl.nf8)); LPVS_TRY(rlo.alloc(sizeof(double) * (size_t)sl.nf8));
        LPVS_TRY(eps.alloc(sizeof(double) * sl.eps.size()));
        LPVS_TRY(copy_to_device(hi.p, sl.om_hi.data(), sizeof(double) * (size_t)sl.nsl, s));
        LPVS_TRY(copy_to_device(lo.p, sl.om_lo.data(), sizeof(double) * (size_t)sl.nsl, s));
        LPVS_TRY(copy_to_device(rhi.p, sl.omr_hi.data(), sizeof(double) * (size_t)sl.nf8, s));
        LPVS_TRY(copy_to_device(rlo.p, sl.omr_lo.data(), sizeof(double) * (size_t)sl.nf8, s));
        LPVS_TRY(copy_to_device(eps.p, sl.eps.data(), sizeof(double) * sl.eps.size(), s));
        return LPVS_OK;
    }
};

// ranges (optional) = {min V, max V, max|V|, max|X|} over ALL rows of the signal when (y, X, V) is only a row shard of it
static int32_t create_lpv_impl(const double *y, int64_t ns, const double *X, const double *V, int64_t N, const double *w, int64_t Nf,
                               int64_t Nv, int32_t normalize, int32_t coulomb, int32_t device, lpvs_problem **out,
                               const double *ranges = nullptr) {
    if (N <= 0 || Nf <= 0 || Nv <= 0 || ns <= 0) { set_error("N, Nf, Nv and the number of signals must be positive"); return LPVS_EARGUMENT; }
    lpvs_problem *h = nullptr;
    LPVS_TRY(problem_begin(device, out, &h));
    struct Guard { lpvs_problem *h; ~Guard() { delete h; } } guard{h};
    hipStream_t s = h->stream;
    const int64_t nb = coulomb ? 2 * Nv : Nv;
    h->kind = 1; h->N = N; h->Nf = Nf; h->nb = nb; h->n = 2 * Nf * nb; h->np = round_up(h->n, 128); h->ns = ns;
    DevArg dy, dX, dV, dw;
    LPVS_TRY(dy.set(y, N * ns, s)); LPVS_TRY(dX.set(X, N, s)); LPVS_TRY(dV.set(V, N, s)); LPVS_TRY(dw.set(w, Nf, s));
    LPVS_TRY(h->G.alloc(sizeof(double) * (size_t)h->np * (size_t)h->np));
    LPVS_TRY(h->b.alloc(sizeof(double) * (size_t)h->np * (size_t)ns));
    LPVS_HIP(hipMemsetAsync(h->G.p, 0, h->G.bytes, s));
    LPVS_HIP(hipMemsetAsync(h->b.p, 0, h->b.bytes, s));
    LPVS_TRY(alloc_state(h));

    // ---- Gram form.  LPVS_GRAM_FORM = auto (default) | ap | krs | kr
    //   ap : w is an arithmetic progression up to rounding -> 3Nf-1 non-uniform Fourier sums per activation pair
    //        (nudft.hip); admitted when max|eps_f| * max|x| <= 1e-7 (second-order term <= 5e-15)
    //   krs / kr : dense MFMA contraction for arbitrary w (gram.hip)
    const int form_opt = option_in_effect(LPVS_OPT_GRAM_FORM, h->opt[LPVS_OPT_GRAM_FORM]);
    const std::string form = form_opt == LPVS_GRAM_AP ? "ap" : form_opt == LPVS_GRAM_KRS ? "krs" : form_opt == LPVS_GRAM_KR ? "kr" : "auto";
    bool use_ap = false;
    ApSlots sl;
    double xam = 0;                                  // max|X| (over all rows of the signal)
    if (form == "auto" || form == "ap") {
        std::vector<double> hw;
        LPVS_TRY(fetch_host(hw, w, Nf));
        double xlo, xhi;
        if (ranges) xam = ranges[3];
        else LPVS_TRY(device_minmax(dX.p, N, &xlo, &xhi, &xam, s));
        sl = make_ap_slots(hw, xam);
        use_ap = sl.ok;
        if (form == "ap" && !use_ap) { set_error("LPVS_GRAM_FORM=ap but w is not an arithmetic progression (max|eps|*max|x| = %.3g)", sl.emax * xam); return LPVS_EARGUMENT; }
    }
    if (use_ap) {
        const int64_t nf8 = sl.nf8, nsl = sl.nsl, P = nb * (nb + 1) / 2;
        const ApStep &step = sl.step;
        double lo, hi, am, gamma; std::vector<double> vc;
        DevBuf K, KK, dvc, part, tab, tabb, nwork, epsr;   // (all temporaries before the drain guard: released only after the stream is idle)
        ApSlotsDev sd;
        DrainOnExit drain(s);
        LPVS_HIP(hipEventRecord(h->ev[0].a, s));
        if (ranges) { lo = ranges[0]; hi = ranges[1]; am = ranges[2]; }
        else LPVS_TRY(device_minmax(dV.p, N, &lo, &hi, &am, s));
        basis_centers(lo, hi, am, Nv, coulomb, vc, &gamma);
        const int64_t ldk = nb + 1;
        LPVS_TRY(dvc.alloc(sizeof(double) * vc.size()));
        LPVS_TRY(copy_to_device(dvc.p, vc.data(), sizeof(double) * vc.size(), s));
        LPVS_TRY(K.alloc(sizeof(double) * (size_t)N * (size_t)ldk));
        LPVS_TRY(KK.alloc(sizeof(double) * (size_t)N * (size_t)P));
        LPVS_TRY(launch_basis_table(dV.p, N, dvc.as<double>(), nb, gamma, normalize, coulomb, K.as<double>(), ldk, s));
        LPVS_TRY(launch_pair_table(K.as<double>(), ldk, nb, N, KK.as<double>(), s));
        LPVS_TRY(sd.upload(sl, s));
        LPVS_HIP(hipEventRecord(h->ev[0].b, s));
        LPVS_TRY(part.alloc(std::max(nudft_partial_bytes(N, nsl, P), nudft_partial_bytes(N, nf8, nb))));
        LPVS_TRY(tab.alloc(sizeof(double) * (size_t)nsl * (size_t)P * 4));
        LPVS_TRY(tabb.alloc(sizeof(double) * (size_t)nf8 * (size_t)nb * 4));
        LPVS_HIP(hipEventRecord(h->ev[1].a, s));
        // merged slot layout: the slot sums are Fourier coefficients at the multiples of ONE step -> non-uniform FFT (nufft.hip);
        // LPVS_NUDFT=direct keeps the direct evaluation of nudft.hip
        const bool nufft_on = option_in_effect(LPVS_OPT_SLOT_SUMS, h->opt[LPVS_OPT_SLOT_SUMS]) != LPVS_SLOTS_DIRECT;
        bool nufft = nufft_on && sl.merged && nufft_applicable(N, nsl, P);
        const int nfg = nufft_grid_size(nsl);
        const double xam_ = xam;
        if (nufft) {
            LPVS_TRY(nwork.alloc(nufft_work_bytes(N, nsl, P)));
            const int32_t rc = launch_nufft_tab(dX.p, nullptr, N, xam_, KK.as<double>(), P, (int)P, step.hi[1], step.lo[1], 0, (int)nsl, nfg, false, nwork.p,
                                                tab.as<double>(), s);
            if (rc == kNufftNonFinite) nufft = false;    // NaN / Inf in the inputs: the direct sums propagate them as the reference does
            else if (rc != LPVS_OK) return rc;
        }
        if (nufft) {
            h->gram_form = 5;
        } else {
            h->gram_form = 4;
            LPVS_TRY(launch_nudft(dX.p, nullptr, N, KK.as<double>(), P, (int)P, sd.hi.as<double>(), sd.lo.as<double>(), (int)nsl, step, part.as<double>(), tab.as<double>(), s));
        }
        LPVS_TRY(launch_ap_assemble(tab.as<double>(), sd.eps.as<double>(), Nf, sl.s0, sl.delta, nb, h->n, h->G.as<double>(), h->np, s));
        LPVS_HIP(hipEventRecord(h->ev[1].b, s));
        LPVS_HIP(hipEventRecord(h->ev[2].a, s));
        // right-hand sides: the slots a + f*D are the modes s0/2 + f of the same step when s0 is even -> same grid, same coordinates
        const bool nufft_rhs = nufft && sl.s0 % 2 == 0 && (int)(sl.s0 / 2 + nf8) <= (int)nsl;
        // mode (s0/2 + f) D is a + f D - delta/2: the residual joins the first-order term (epsr)
        if (nufft_rhs) {
            std::vector<double> er(sl.eps.size());
            for (size_t f = 0; f < er.size(); ++f) er[f] = sl.eps[f] + 0.5 * sl.delta;
            LPVS_TRY(epsr.alloc(sizeof(double) * er.size()));
            LPVS_TRY(copy_to_device(epsr.p, er.data(), sizeof(double) * er.size(), s));
            LPVS_HIP(hipStreamSynchronize(s));       // (er is a local)
        }
        for (int64_t q = 0; q < ns; ++q) {   // b_q = Phi' y_q: Nf slots a + f*D, weights y K_j, first-order eps correction
            bool by_nufft = nufft_rhs;
            if (by_nufft) {
                const int32_t rc = launch_nufft_tab(dX.p, dy.p + q * N, N, xam_, K.as<double>(), ldk, (int)nb, step.hi[1], step.lo[1], (int)(sl.s0 / 2), (int)nf8, nfg,
                                                    true, nwork.p, tabb.as<double>(), s);
                if (rc == kNufftNonFinite) by_nufft = false;   // a NaN / Inf in this signal
                else if (rc != LPVS_OK) return rc;
            }
            if (!by_nufft)
                LPVS_TRY(launch_nudft(dX.p, dy.p + q * N, N, K.as<double>(), ldk, (int)nb, sd.rhi.as<double>(), sd.rlo.as<double>(), (int)nf8, step, part.as<double>(), tabb.as<double>(), s));
            LPVS_TRY(launch_ap_rhs(tabb.as<double>(), by_nufft ? epsr.as<double>() : sd.eps.as<double>(), Nf, nb, h->b.as<double>() + q * h->np, s));
        }
        LPVS_HIP(hipEventRecord(h->ev[2].b, s));
        LPVS_HIP(hipStreamSynchronize(s));
        h->t_basis = h->ev[0].ms(); h->t_gram = h->ev[1].ms(); h->t_reduce = h->ev[2].ms();
        h->gram_launches = 8.0 * (double)N * (double)nsl * (double)P;   // flops the structured form issues (4 fma per sample, slot, pair)
        h->gram_flops = (double)N * (double)h->n * (double)(h->n + 1);
        guard.h = nullptr;
        *out = h;
        return LPVS_OK;
    }

    const bool krs = form != "kr" && gram_krs_fits(nb);   // symmetric-pair form unless the pair table does not fit LDS
    const GramPlan pl = krs ? make_gram_plan_pairs(Nf, nb, N) : make_gram_plan(h->n, N);
    const int64_t Npad = pl.ksplit * pl.rows_per_chunk;

    DevBuf T, K, KK, G3, slab, scr; int64_t ldk;
    DrainOnExit drain(s);
    LPVS_HIP(hipEventRecord(h->ev[0].a, s));
    LPVS_TRY(build_lpv_tables(dX.p, dV.p, N, Npad, dw.p, Nf, Nv, normalize, coulomb, T, K, &ldk, s, ranges));
    if (krs) {
        LPVS_TRY(KK.alloc(sizeof(double) * (size_t)Npad * (size_t)pl.pairs));
        LPVS_TRY(launch_pair_table(K.as<double>(), ldk, nb, Npad, KK.as<double>(), s));
    }
    LPVS_HIP(hipEventRecord(h->ev[0].b, s));
    LPVS_TRY(slab.alloc(pl.slab_bytes));
    LPVS_TRY(scr.alloc(rhs_scratch_bytes(N, h->n)));
    if (krs) LPVS_TRY(G3.alloc(sizeof(double) * (size_t)pl.np2 * (size_t)(pl.np2 * pl.pairs)));
    // b_q = Phi' y_q is HBM-bound (it streams the trig table) while the Gram is MFMA-bound and leaves registers
    // free for small workgroups: run it on a side stream underneath the Gram kernel.
    struct Side { hipStream_t s = nullptr; hipEvent_t ready = nullptr, done = nullptr;
                  ~Side() { if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } if (ready) (void)hipEventDestroy(ready); if (done) (void)hipEventDestroy(done); } } side;
    LPVS_HIP(hipStreamCreateWithFlags(&side.s, hipStreamNonBlocking));
    LPVS_HIP(hipEventCreateWithFlags(&side.ready, hipEventDisableTiming));
    LPVS_HIP(hipEventCreateWithFlags(&side.done, hipEventDisableTiming));
    LPVS_HIP(hipEventRecord(side.ready, s));                 // tables are complete at this point of the main stream
    LPVS_HIP(hipStreamWaitEvent(side.s, side.ready, 0));
    LPVS_HIP(hipEventRecord(h->ev[1].a, s));
    if (krs) LPVS_TRY(launch_gram_krs(pl, T.as<double2>(), Nf, KK.as<double>(), nb, slab.as<double>(), s));
    else LPVS_TRY(launch_gram_kr(pl, T.as<double2>(), Nf, K.as<double>(), ldk, nb, slab.as<double>(), s));
    LPVS_HIP(hipEventRecord(h->ev[1].b, s));
    for (int64_t q = 0; q < ns; ++q)   // b_q = Phi' y_q for every signal sharing the regressor
        LPVS_TRY(launch_rhs_kr(T.as<double2>(), Nf, K.as<double>(), ldk, nb, dy.p + q * N, N, h->b.as<double>() + q * h->np, scr.as<double>(), scr.bytes, side.s));
    LPVS_HIP(hipEventRecord(side.done, side.s));
    LPVS_HIP(hipEventRecord(h->ev[2].a, s));
    if (krs) LPVS_TRY(launch_gram_reduce_krs(pl, slab.as<double>(), nb, G3.as<double>(), h->G.as<double>(), h->np, s));
    else LPVS_TRY(launch_gram_reduce(pl, slab.as<double>(), h->G.as<double>(), h->np, s));
    LPVS_HIP(hipStreamWaitEvent(s, side.done, 0));
    LPVS_HIP(hipEventRecord(h->ev[2].b, s));
    LPVS_HIP(hipStreamSynchronize(s));
    h->t_basis = h->ev[0].ms(); h->t_gram = h->ev[1].ms(); h->t_reduce = h->ev[2].ms();
    h->gram_launches = (double)pl.tiles * 128.0 * 256.0 * 2.0 * (double)(pl.ksplit * pl.rows_per_chunk); h->gram_flops = (double)N * (double)h->n * (double)(h->n + 1);
    h->gram_form = krs ? 2 : 1;
    guard.h = nullptr;
    *out = h;
    return LPVS_OK;
}

int32_t lpvs_problem_create_lpv_f64(const double *y, const double *X, const double *V, int64_t N, const double *w, int64_t Nf,
                                    int64_t Nv, int32_t normalize, int32_t coulomb, int32_t device, lpvs_problem **out) {
    return create_lpv_impl(y, 1, X, V, N, w, Nf, Nv, normalize, coulomb, device, out);
}

int32_t lpvs_problem_create_lpv_multi_f64(const double *Y, int64_t ns, const double *X, const double *V, int64_t N, const double *w,
                                          int64_t Nf, int64_t Nv, int32_t normalize, int32_t coulomb, int32_t device,
                                          lpvs_problem **out) {
    return create_lpv_impl(Y, ns, X, V, N, w, Nf, Nv, normalize, coulomb, device, out);
}

int32_t lpvs_lpv_ranges_f64(const double *X, const double *V, int64_t N, double *out4) {
    if (!X || !V || !out4 || N <= 0) { set_error("NULL argument or N <= 0"); return LPVS_EARGUMENT; }
    if (lpvs_device_count() == 0) { set_error("no HIP device visible (the gfx950 path has no CPU fallback)"); return LPVS_EDEVICE; }
    hipStream_t s = nullptr;
    DevArg dX, dV;
    LPVS_TRY(dX.set(X, N, s)); LPVS_TRY(dV.set(V, N, s));
    double xlo, xhi;
    LPVS_TRY(device_minmax(dV.p, N, &out4[0], &out4[1], &out4[2], s));
    LPVS_TRY(device_minmax(dX.p, N, &xlo, &xhi, &out4[3], s));
    return LPVS_OK;
}

int32_t lpvs_problem_create_lpv_rows_f64(const double *Y, int64_t ns, const double *X, const double *V, int64_t N_local, const double *w,
                                         int64_t Nf, int64_t Nv, int32_t normalize, int32_t coulomb, const double *ranges4,
                                         int32_t device, lpvs_problem **out) {
    if (!ranges4) { set_error("ranges4 is NULL"); return LPVS_EARGUMENT; }
    if (!(ranges4[0] <= ranges4[1]) || !(ranges4[2] >= 0) || !(ranges4[3] >= 0)) { set_error("ranges4 = {min V, max V, max|V|, max|X|} is inconsistent"); return LPVS_EARGUMENT; }
    return create_lpv_impl(Y, ns, X, V, N_local, w, Nf, Nv, normalize, coulomb, device, out, ranges4);
}

int32_t lpvs_problem_device_gram_f64(lpvs_problem *h, double **G_dev, double **b_dev, int64_t *np) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    if (G_dev) *G_dev = h->G.as<double>();
    if (b_dev) *b_dev = h->b.as<double>();
    if (np) *np = h->np;
    return LPVS_OK;
}

int32_t lpvs_problem_gram_modified(lpvs_problem *h) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    h->M_valid = false; h->Mp_valid = false; h->inited = false;
    h->drop_graph();
    return LPVS_OK;
}

int32_t lpvs_problem_create_fourier_f64(const double *y, const double *t, int64_t N, const double *f, int64_t Nf, const double *W,
                                        int32_t device, lpvs_problem **out) {
    if (N <= 0 || Nf <= 0) { set_error("N and Nf must be positive"); return LPVS_EARGUMENT; }
    int64_t zf = 0;
    LPVS_TRY(lpvs_check_freq_f64(f, Nf, &zf));
    lpvs_problem *h = nullptr;
    LPVS_TRY(problem_begin(device, out, &h));
    struct Guard { lpvs_problem *h; ~Guard() { delete h; } } guard{h};
    hipStream_t s = h->stream;
    h->kind = 0; h->N = N; h->Nf = Nf; h->zerofreq = zf; h->n = zf ? 2 * Nf - 1 : 2 * Nf;
    DevArg dt, df;
    LPVS_TRY(dt.set(t, N, s)); LPVS_TRY(df.set(f, Nf, s));
    // structured Gram when T(2pi)*f is an arithmetic progression (default_freqs and every grid of the reference's tests)
    const int form_opt = option_in_effect(LPVS_OPT_GRAM_FORM, h->opt[LPVS_OPT_GRAM_FORM]);
    const std::string form = form_opt == LPVS_GRAM_AP ? "ap" : form_opt == LPVS_GRAM_KRS ? "krs" : form_opt == LPVS_GRAM_KR ? "kr" : "auto";
    if (form == "auto" || form == "ap") {
        std::vector<double> hw;
        LPVS_TRY(fetch_host(hw, f, Nf));
        for (auto &v : hw) v = 6.283185307179586 * v;          // rounded as the regressor kernel does (src/lsfft.jl:33)
        double tlo, thi, tam;
        LPVS_TRY(device_minmax(dt.p, N, &tlo, &thi, &tam, s));
        const ApSlots sl = make_ap_slots(hw, tam);
        if (form == "ap" && !sl.ok) { set_error("LPVS_GRAM_FORM=ap but 2*pi*f is not an arithmetic progression (max|eps|*max|t| = %.3g)", sl.emax * tam); return LPVS_EARGUMENT; }
        if (sl.ok) {
            h->np = round_up(h->n, 128);
            DevArg dy, dW;
            LPVS_TRY(dy.set(y, N, s));
            if (W) LPVS_TRY(dW.set(W, N, s));
            LPVS_TRY(h->G.alloc(sizeof(double) * (size_t)h->np * (size_t)h->np));
            LPVS_TRY(h->b.alloc(sizeof(double) * (size_t)h->np));
            LPVS_HIP(hipMemsetAsync(h->G.p, 0, h->G.bytes, s));
            LPVS_HIP(hipMemsetAsync(h->b.p, 0, h->b.bytes, s));
            LPVS_TRY(alloc_state(h));
            ApSlotsDev sd; DevBuf part, tab, tabb;
            DrainOnExit drain(s);
            LPVS_HIP(hipEventRecord(h->ev[0].a, s));
            LPVS_TRY(sd.upload(sl, s));
            LPVS_HIP(hipEventRecord(h->ev[0].b, s));
            // (BOTH launches below share `part`: the right-hand side's has a third of the slots but -- fewer slot groups, so smaller chunks --
            // up to four times the chunks.  Sized for the Gram launch alone it overflowed for N / rows-per-chunk in [171, 256), e.g.
            // N = 118727 or 204800: found as a GPU fault in test_fourier_gram_additivity_cfg2 when the minimum chunk went from 512 to 64.)
            LPVS_TRY(part.alloc(std::max(nudft_partial_bytes(N, sl.nsl, 1), nudft_partial_bytes(N, sl.nf8, 1))));
            LPVS_TRY(tab.alloc(sizeof(double) * (size_t)sl.nsl * 4));
            LPVS_TRY(tabb.alloc(sizeof(double) * (size_t)sl.nf8 * 4));
            const double *Wd = W ? dW.p : nullptr;               // A' diag(W) A and A' (W .* y), src/lasso.jl:119-120
            LPVS_HIP(hipEventRecord(h->ev[1].a, s));
            LPVS_TRY(launch_nudft(dt.p, nullptr, N, Wd, 1, 1, sd.hi.as<double>(), sd.lo.as<double>(), (int)sl.nsl, sl.step, part.as<double>(), tab.as<double>(), s));
            LPVS_TRY(launch_ap_assemble_fourier(tab.as<double>(), sd.eps.as<double>(), Nf, sl.s0, sl.delta, (int)zf, h->n, h->G.as<double>(), h->np, 1, 0, 0, s));
            LPVS_HIP(hipEventRecord(h->ev[1].b, s));
            LPVS_HIP(hipEventRecord(h->ev[2].a, s));
            LPVS_TRY(launch_nudft(dt.p, dy.p, N, Wd, 1, 1, sd.rhi.as<double>(), sd.rlo.as<double>(), (int)sl.nf8, sl.step, part.as<double>(), tabb.as<double>(), s));
            LPVS_TRY(launch_ap_rhs_fourier(tabb.as<double>(), sd.eps.as<double>(), Nf, (int)zf, h->b.as<double>(), 1, 0, 0, s));
            LPVS_HIP(hipEventRecord(h->ev[2].b, s));
            LPVS_HIP(hipStreamSynchronize(s));
            h->t_basis = h->ev[0].ms(); h->t_gram = h->ev[1].ms(); h->t_reduce = h->ev[2].ms();
            h->gram_launches = 8.0 * (double)N * (double)sl.nsl;
            h->gram_flops = (double)N * (double)h->n * (double)(h->n + 1);
            h->gram_form = 4;
            guard.h = nullptr;
            *out = h;
            return LPVS_OK;
        }
    }
    LPVS_TRY(create_panel_problem(h, y, W, N, [&](double *P, int64_t ld) {
        return launch_fourier_panel(dt.p, N, df.p, Nf, (int)zf, P, ld, s);
    }));
    guard.h = nullptr;
    *out = h;
    return LPVS_OK;
}

int32_t lpvs_problem_create_dense_f64(const double *A, const double *y, int64_t m, int64_t n, const double *W, int32_t device,
                                      lpvs_problem **out) {
    if (m <= 0 || n <= 0) { set_error("m and n must be positive"); return LPVS_EARGUMENT; }
    lpvs_problem *h = nullptr;
    LPVS_TRY(problem_begin(device, out, &h));
    struct Guard { lpvs_problem *h; ~Guard() { delete h; } } guard{h};
    hipStream_t s = h->stream;
    h->kind = 2; h->N = m; h->n = n;
    DevArg dA;
    LPVS_TRY(dA.set(A, m * n, s));
    LPVS_TRY(create_panel_problem(h, y, W, m, [&](double *P, int64_t ld) {
        return launch_transpose_to_panel(dA.p, m, n, P, ld, s);
    }));
    guard.h = nullptr;
    *out = h;
    return LPVS_OK;
}

int32_t lpvs_problem_create_gram_f64(const double *G, const double *b, int64_t n, int32_t device, lpvs_problem **out) {
    if (n <= 0) { set_error("n must be positive"); return LPVS_EARGUMENT; }
    lpvs_problem *h = nullptr;
    LPVS_TRY(problem_begin(device, out, &h));
    struct Guard { lpvs_problem *h; ~Guard() { delete h; } } guard{h};
    hipStream_t s = h->stream;
    h->kind = 2; h->n = n; h->np = round_up(n, 128);
    LPVS_TRY(h->G.alloc(sizeof(double) * (size_t)h->np * (size_t)h->np));
    LPVS_TRY(h->b.alloc(sizeof(double) * (size_t)h->np));
    LPVS_HIP(hipMemsetAsync(h->G.p, 0, h->G.bytes, s));
    LPVS_HIP(hipMemsetAsync(h->b.p, 0, h->b.bytes, s));
    LPVS_TRY(alloc_state(h));
    LPVS_HIP(hipMemcpy2DAsync(h->G.p, sizeof(double) * (size_t)h->np, G, sizeof(double) * (size_t)n, sizeof(double) * (size_t)n, (size_t)n,
                              is_device_ptr(G) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    LPVS_HIP(hipStreamSynchronize(s));
    LPVS_TRY(copy_to_device(h->b.p, b, sizeof(double) * (size_t)n, s));
    guard.h = nullptr;
    *out = h;
    return LPVS_OK;
}

int32_t lpvs_problem_destroy(lpvs_problem *h) {
    if (h == nullptr) return LPVS_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    delete h;
    return LPVS_OK;
}

int32_t lpvs_problem_size(const lpvs_problem *h, int64_t *n) {
    if (!h || !n) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    *n = h->n;
    return LPVS_OK;
}
int32_t lpvs_problem_zerofreq(const lpvs_problem *h, int64_t *zerofreq) {
    if (!h || !zerofreq) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    *zerofreq = h->zerofreq;
    return LPVS_OK;
}

int32_t lpvs_problem_get_gram_f64(lpvs_problem *h, double *G_out, double *b_out) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    if (G_out) {
        LPVS_HIP(hipMemcpy2DAsync(G_out, sizeof(double) * (size_t)h->n, h->G.p, sizeof(double) * (size_t)h->np, sizeof(double) * (size_t)h->n,
                                  (size_t)h->n, is_device_ptr(G_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
        LPVS_HIP(hipStreamSynchronize(h->stream));
    }
    if (b_out) LPVS_TRY(copy_from_device(b_out, h->b.p, sizeof(double) * (size_t)h->n, h->stream));
    return LPVS_OK;
}

int32_t lpvs_problem_get_rhs_f64(lpvs_problem *h, double *b_out) {
    if (!h || !b_out) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    return copy_state_out(h, h->b.p, b_out);
}

int32_t lpvs_problem_get_inverse_f64(lpvs_problem *h, double shift, double *Minv_out) {
    if (!h || !Minv_out) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    h->inited = false;   // M is re-used
    LPVS_TRY(factorize(h, shift));
    LPVS_HIP(hipMemcpy2DAsync(Minv_out, sizeof(double) * (size_t)h->n, h->M.p, sizeof(double) * (size_t)h->np, sizeof(double) * (size_t)h->n,
                              (size_t)h->n, is_device_ptr(Minv_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    LPVS_HIP(hipStreamSynchronize(h->stream));
    return LPVS_OK;
}

int32_t lpvs_problem_solve_ridge_f64(lpvs_problem *h, double ridge, double *x_out) {
    if (!h || !x_out) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    h->inited = false;  // M is re-used
    LPVS_TRY(factorize(h, ridge));
    // two rounds of iterative refinement against G + ridge I: the normal equations square cond(A) and the reference solves
    // the same systems by SVD / QR (src/utilities.jl:49-60), so the explicit inverse alone would lose cond(G) * eps
    LPVS_TRY(launch_ridge_solve_refined(h->G.as<double>(), h->M.as<double>(), h->np, h->n, h->b.as<double>(), ridge, 2,
                                        h->scratch.as<double>(), h->rhs.as<double>(), h->z.as<double>(), h->stream));
    // fail loudly instead of returning a meaningless solution when G + ridge I is singular to working precision (the
    // reference's QR / SVD of [A; lam I] still works there: the host wrapper then takes that route)
    std::vector<double> hr((size_t)h->n), hb((size_t)h->n);
    LPVS_TRY(copy_from_device(hr.data(), h->z.p, sizeof(double) * (size_t)h->n, h->stream));
    LPVS_TRY(copy_from_device(hb.data(), h->b.p, sizeof(double) * (size_t)h->n, h->stream));
    double r2 = 0, b2 = 0;
    for (int64_t i = 0; i < h->n; ++i) { r2 += hr[i] * hr[i]; b2 += hb[i] * hb[i]; }
    if (!(r2 <= 1e-18 * b2) && b2 > 0) {
        h->M_valid = false;
        set_error("normal equations (G + %.3g I) x = b are too ill-conditioned for the device solve (relative residual %.3g after refinement)",
                  ridge, std::sqrt(r2 / b2));
        return LPVS_ENUMERIC;
    }
    return copy_from_device(x_out, h->scratch.p, sizeof(double) * (size_t)h->n, h->stream);
}

int32_t lpvs_problem_set_prox(lpvs_problem *h, int32_t kind, double param, int64_t group_len) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    if (kind < LPVS_PROX_L1 || kind > LPVS_PROX_GROUP_L2) { set_error("unknown prox kind %d", kind); return LPVS_EUNSUPPORTED; }
    if (kind == LPVS_PROX_GROUP_L2 && group_len <= 0) { set_error("group_len must be positive"); return LPVS_EARGUMENT; }
    if (kind == LPVS_PROX_GROUP_L2 && group_len > 8192) { set_error("group_len > 8192 is not supported by the device prox"); return LPVS_EUNSUPPORTED; }
    h->prox_kind = kind; h->prox_param = param; h->group_len = group_len;
    h->drop_graph();   // kernel parameters are baked into the captured graph
    return LPVS_OK;
}

int32_t lpvs_admm_init_f64(lpvs_problem *h, const double *x0, double mu, double tol, int32_t linear_sign) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    if (!(mu >= 0 && mu <= 1)) { set_error("μ should be ≤ 1"); return LPVS_EASSERT; }  // src/lasso.jl:143
    if (mu == 0) { set_error("mu = 0 makes the x-update singular"); return LPVS_ENUMERIC; }
    if (linear_sign != 1 && linear_sign != -1) { set_error("linear_sign must be +1 or -1"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    h->drop_graph();
    LPVS_TRY(factorize(h, 1.0 / mu));            // no-op when M is cached for this shift; clears Mp_valid otherwise
    // ---- the x-update correction (admm.hip, launch_xupdate_correction): decided first, the packing below depends on it.
    // Default: every handle of n >= 2048, after the iterations 16, 512, 1024, 2048, ... (three corrections in 2000 iterations).
    // Round 6: handles with several right-hand sides too -- at cfg5's judged size (n = 32768, 8 channels) the uncorrected run drifts 4.1e-10
    // in x, z (1.4e-10 in u) from the corrected one after 2000 iterations (profiles/r06_cfg5_xcorr_fullsize.txt), the same constant forcing
    // E w as at cfg3; a correction there is an accurate product over the 8.6-GB f64 Gram for all channels in one pass: 8.6 ms, three of them
    // 1.9 % of a 2000-iteration run.  LPVS_XCORR_OFF switches it off per handle.
    // LPVS_XUPDATE_CORRECTION (A/B measurements): "0" none; "B" after B^j; "eN" after 16 and every N-th; "dN" after 16, N, 2N, 4N, ...
    const bool offset_form_wanted = h->np >= kSymmetricMinNp;
    const int xc_opt = option_in_effect(LPVS_OPT_XUPDATE_CORRECTION, h->opt[LPVS_OPT_XUPDATE_CORRECTION]);   // explicit, thread default, or environment ("0" = off)
    const bool xc_on = offset_form_wanted && xc_opt != LPVS_XCORR_OFF;
    // Schedule (round 6): after 16, then 128 and its doublings for one right-hand side (five corrections in 2000 iterations: +0.4 ms at cfg3) --
    // the off-family sweep (tests/test_gpu_offfamily.py, profiles/r06_offfamily_probe*.txt) has cases at cond(G + I/mu) ~ 2e4 where 16, 512, ...
    // leaves x, z 8e-10 from the exact iterates after 600 iterations and this schedule 1.5e-10; 512 and its doublings for several right-hand sides
    // (a correction there is a pass over the whole f64 Gram for all channels: 8.6 ms at cfg5; measured 2.6e-10 from the CPU reference iterates at n = 8192: tests/test_gpu_configs.py).
    h->xcorr_base = 0; h->xcorr_every = xc_on ? (h->ns == 1 ? 128 : 512) : 0; h->xcorr_double = true;
    const char *xc_env = xc_on ? experiment_env("LPVS_XUPDATE_CORRECTION") : nullptr;   // (schedules: an experiment knob; "0" = off is the option's plain fallback, option_from_env)
    if (xc_env != nullptr && xc_env[0] == '0' && xc_env[1] == 0) xc_env = nullptr;   // ("0" only says off, and only as the option's last fallback)
    if (const char *e = xc_env) {                                                    // schedule experiments: "B", "eN", "dN", "qN"
        const bool sched = (e[0] == 'e' || e[0] == 'd' || e[0] == 'q') && atoi(e + 1) > 0;
        const bool base = e[0] >= '0' && e[0] <= '9' && atoi(e) >= 2;
        if (sched) {
            h->xcorr_double = e[0] == 'd' || e[0] == 'q'; h->xcorr_early = e[0] == 'q';   // "qN": after 1, 2, 4, 8, 16, N, 2N, 4N, ...
            h->xcorr_base = 0; h->xcorr_every = atoi(e + 1);
        } else if (base) { h->xcorr_every = 0; h->xcorr_base = atoi(e); }
        // anything else ("1", "on", "true", ...) only says ON: the default schedule set above stays (it used to fall through
        // to base = 0, i.e. silently OFF, and took the 32-bit reads with it)
    }
    // Corrected single-signal handles READ 32 bits of the 36 their fixed-point tiles hold (4 B per element instead of 4.5) and carry the
    // product of the 4-bit planes with a right-hand side up to nib_period iterations old in the offset vector (admm.hip, "the stale nibble
    // product"): x, z and u all stay where the 36-bit reads leave them (u within 5.0e-10 of the exact iterates at cfg3; plain truncation to
    // 32 bits: 2e-9 .. 7e-9 -- the dual variable integrates it; profiles/r05_cfg3_stale_nibble_product.txt, r05_cfg3_fixbits.txt).
    // The default, and LPVS_STORAGE_MIXED32 by name; LPVS_STORAGE_MIXED asks for the 36-bit reads.
    const int st_asked = option_in_effect(LPVS_OPT_M_STORAGE, h->opt[LPVS_OPT_M_STORAGE]);
    const bool read32 = h->ns == 1 && !h->f32 && h->xcorr() && (st_asked == 0 || st_asked == LPVS_STORAGE_MIXED32);
    int fix_bits = 36;                               // (packed with all 36 bits: the nibble planes feed the stale nibble product)
    if (const char *e = experiment_env("LPVS_FIX_BITS")) fix_bits = atoi(e) >= 20 && atoi(e) <= 36 ? atoi(e) : fix_bits;
    h->nib_period = read32 ? 32 : 0;                 // LPVS_NIB_PERIOD: refresh period of the stale nibble product (experiments; 0: no stale product, the 36-bit reads)
    if (const char *e = read32 ? experiment_env("LPVS_NIB_PERIOD") : nullptr) h->nib_period = atoi(e) > 0 ? atoi(e) : 0;
    // ... denser while the right-hand side still moves fast: after every launch up to 15, every 2nd up to 31, 4th up to 63, ..., 32nd from 256 on
    // (103 refreshes in 2000 iterations; u at cfg3 after 200 iterations 1.75e-10 from the exact iterate instead of 3.9e-10 -- 36-bit reads: 1.40e-10)
    h->nib_ramp = read32 ? 8 : 0;
    if (const char *e = read32 ? experiment_env("LPVS_NIB_RAMP") : nullptr) h->nib_ramp = atoi(e) > 0 ? atoi(e) : 0;
    h->Mp_read32 = read32;
    if (h->Mp_valid && h->Mp_mode == kMpMixed && h->Mp_fix_bits != fix_bits) h->Mp_valid = false;   // (the same M packed for the other choice)
    const int mode = mp_mode_for(h);
    const bool demoted = mode == kMpMixed && h->Mp_mode == kMpSplit && h->Mp_demoted;   // mixed was tried for this M and found no small tiles
    if (h->np >= kSymmetricMinNp && (!h->Mp_valid || (h->Mp_mode != mode && !demoted))) {   // tile-packed lower triangle for the half-traffic mat-vec
        h->Mp_demoted = false;
        bool just_demoted = false;   // set only by the demotion a few lines down: the handle's OLD (Mp_mode, Mp_valid) must not be mistaken for it
        const size_t elt = mode == kMpF32 ? 4 : (mode == kMpSplit || mode == kMpMixed ? 6 : 8);
        const size_t ntiles = symv_packed_doubles(h->np) / (128 * 128);
        const size_t need = elt * symv_packed_doubles(h->np) + (mode == kMpMixed ? ((ntiles + 255) / 256) * 256 + 256 : 0);   // + tile types + max|M|
        if (!h->Mp.p || h->Mp.bytes < need) LPVS_TRY(h->Mp.alloc(need));
        h->Mp_stream_bytes = (double)elt * (double)symv_packed_doubles(h->np);
        h->Mp_fixed_tiles = 0; h->Mp_fixed_diag = 0;
        if (mode == kMpF32) LPVS_TRY(launch_pack_tiles_f32(h->M.as<double>(), h->np, h->Mp.as<float>(), s));
        else if (mode == kMpSplit) LPVS_TRY(launch_pack_tiles_split(h->M.as<double>(), h->np, h->Mp.as<unsigned char>(), s));
        else if (mode == kMpMixed) {
            unsigned char *types = h->Mp.as<unsigned char>() + 6 * symv_packed_doubles(h->np);
            // (single-signal handles: the packing pass also leaves the largest absolute row sum of M, which the one-launch iteration's quantum
            // bound needs -- one more pass over the matrix otherwise; the tile partials' buffer and the scratch vector are free here)
            unsigned long long *amax = reinterpret_cast<unsigned long long *>(types + ((ntiles + 255) / 256) * 256);
            const bool want_R = h->ns == 1;
            LPVS_TRY(launch_pack_tiles_mixed(h->M.as<double>(), h->np, h->Mp.as<unsigned char>(), types, amax, s, /*diag_float=*/h->ns > 1,
                                             want_R ? h->part.as<double>() : nullptr, h->n, want_R ? h->scratch.as<double>() : nullptr, fix_bits));
            h->Mp_fix_bits = fix_bits;
            std::vector<unsigned char> ht(ntiles);
            unsigned long long rbits = 0;
            LPVS_HIP(hipMemcpyAsync(ht.data(), types, ntiles, hipMemcpyDeviceToHost, s));
            if (want_R) LPVS_HIP(hipMemcpyAsync(&rbits, amax + 1, sizeof(rbits), hipMemcpyDeviceToHost, s));
            LPVS_HIP(hipStreamSynchronize(s));
            if (want_R) memcpy(&h->Mp_rowsum, &rbits, sizeof(double));
            size_t ndiag = 0;
            for (unsigned char t : ht) { h->Mp_fixed_tiles += t != 0; ndiag += t == 2; }
            h->Mp_fixed_diag = (int64_t)ndiag;
            h->Mp_stream_bytes = (double)h->Mp_fixed_tiles * (double)kMixedFixedTileBytes + (double)ndiag * 1024.0 +
                                 (double)(ntiles - (size_t)h->Mp_fixed_tiles) * (double)kMixedFloatTileBytes;
            const size_t nblk_ = (size_t)(h->np / 128);
            if (getenv("LPVS_TRACE")) fprintf(stderr, "[lpvs] mixed packing: %lld of %zu tiles fixed point (%zu diagonal), ns = %lld\n", (long long)h->Mp_fixed_tiles, ntiles, nblk_, (long long)h->ns);
            if (2 * (size_t)h->Mp_fixed_tiles < ntiles) {
                // not a diagonally dominant inverse: the mixed kernel (three workgroups per CU, float-head tiles in two halves) would
                // only lose against the plain 6-byte kernel -- store every tile in the float-head format
                LPVS_TRY(launch_pack_tiles_split(h->M.as<double>(), h->np, h->Mp.as<unsigned char>(), s));
                h->Mp_fixed_tiles = 0; h->Mp_fixed_diag = 0;
                h->Mp_stream_bytes = 6.0 * (double)symv_packed_doubles(h->np);
                h->Mp_valid = true; h->Mp_mode = kMpSplit; h->Mp_demoted = true; just_demoted = true;
            }
        } else LPVS_TRY(launch_pack_tiles(h->M.as<double>(), h->np, h->Mp.as<double>(), s));
        if (!just_demoted) { h->Mp_valid = true; h->Mp_mode = mode; }
    }
    LPVS_HIP(hipMemsetAsync(h->part.p, 0, h->part.bytes, s));   // zero the ticket / block norms
    h->mu = mu; h->tol = tol; h->sign = linear_sign;
    const size_t v = sizeof(double) * (size_t)h->np * (size_t)h->ns;
    LPVS_HIP(hipMemsetAsync(h->x.p, 0, v, s));
    if (x0) LPVS_TRY(copy_state_in(h, h->x.p, x0));
    // signed linear term
    std::vector<double> hb((size_t)h->np * (size_t)h->ns, 0.0);
    LPVS_HIP(hipMemcpyAsync(hb.data(), h->b.p, v, hipMemcpyDeviceToHost, s));
    LPVS_HIP(hipStreamSynchronize(s));
    if (linear_sign < 0) for (auto &q : hb) q = -q;
    LPVS_TRY(copy_to_device(h->bs.p, hb.data(), v, s));
    // reduced-precision copies of M (split, f32) are only ever applied to (z-u)/mu: x = xb + M~ (z-u)/mu, xb = M b in full precision
    // (round 5: the 8-byte storage too -- the offset vector is refined against a double-double residual, which the in-loop product M (b + v) cannot be)
    h->offset_form = h->np >= kSymmetricMinNp;
    if (h->offset_form) {
        if (!h->xb.p) LPVS_TRY(h->xb.alloc(v));
        // every signal's M b, refined once against the Gram the handle still holds, residual in twice the mantissa (admm.hip).  Round 5 left
        // that to the first correction for corrected handles ("sixteen iterations with an offset vector good to ~1e-12 do not show") -- they do
        // show off the benchmark's input family: at cond(G + I/mu) = 2.8e5 (mu = 1, unnormalised basis) the sixteen x-updates with E b in them
        // are integrated by the dual variable and stay there (decay 1 - 1/(mu g) per iteration): u 9.0e-9 from the exact iterates after 600
        // iterations against 3.0e-10 with the offset vector refined here (profiles/r06_offfamily_probe_refine.txt).  One accurate product: 0.2 ms
        // at n = 8192.  rhs and scratch are free until launch_admm_init below writes the state.  LPVS_XB_REFINE = rounds (A/B measurements)
        int steps = 1;
        if (const char *e = experiment_env("LPVS_XB_REFINE")) steps = atoi(e) < 0 ? 0 : atoi(e);
        h->xb_refined = steps > 0;
        LPVS_TRY(launch_offset_vector_refined(h->G.as<double>(), h->M.as<double>(), h->np, h->n, (int)h->ns, h->bs.as<double>(), h->M_shift, steps,
                                              h->xb.as<double>(), h->rhs.as<double>(), h->scratch.as<double>(), s));
        if (h->xcorr()) {
            if (!h->xb0.p) LPVS_TRY(h->xb0.alloc(v));
            if (!h->corr.p) LPVS_TRY(h->corr.alloc(3 * v));
            LPVS_HIP(hipMemcpyAsync(h->xb0.p, h->xb.p, v, hipMemcpyDeviceToDevice, s));
        }
        if (h->nib_period > 0 && h->Mp_mode == kMpMixed) {   // the offset vector without its nibble term: none yet (the first refresh follows launch 1)
            if (!h->xb_corr.p) LPVS_TRY(h->xb_corr.alloc(v));
            if (!h->nib_rhs.p) LPVS_TRY(h->nib_rhs.alloc(v));
            if (!h->nib_acc.p) LPVS_TRY(h->nib_acc.alloc(v));
            LPVS_HIP(hipMemsetAsync(h->nib_acc.p, 0, v, s));
            { const size_t nb = (size_t)(h->np / 128), nt = nb * (nb + 1) / 2; if (!h->nib_part.p) LPVS_TRY(h->nib_part.alloc(sizeof(double) * 2 * nt * 128)); }
            LPVS_HIP(hipMemcpyAsync(h->xb_corr.p, h->xb.p, v, hipMemcpyDeviceToDevice, s));
        } else h->nib_period = 0;
    } else { h->xcorr_base = 0; h->xcorr_every = 0; }
    h->k_enq = 0;
    if (h->offset_form && h->ns == 1 && (h->Mp_mode == kMpMixed || h->Mp_mode == kMpF32) && !h->fi.p) LPVS_TRY(h->fi.alloc(sizeof(double) * fi_doubles(h->np)));
    const AdmmParams p = make_params(h);
    LPVS_TRY(launch_admm_init(p, s));
    h->fi_sync = -1;
    if (p.fi) {   // constants (largest row sum of M, max|xb|) and the records of iteration 0
        if (h->Mp_rowsum > 0 && h->Mp_mode == kMpMixed) {
            AdmmParams pr = p; pr.fi_R = h->Mp_rowsum;
            LPVS_TRY(launch_fi_setup(pr, 0, true, s, h->Mp_rowsum));
        } else LPVS_TRY(launch_fi_setup(p, 0, true, s));
        double hc[2] = {0, 0};
        LPVS_TRY(fi_read_consts(p, hc, s));              // (synchronises)
        h->fi_R = hc[0]; h->fi_xbmax = hc[1] * (1.0 + 0x1p-20); h->fi_sync = 0;   // (max|xb| bounds the corrected offset vector too: it moves by ~1e-12 of it)
    }
    LPVS_HIP(hipStreamSynchronize(s));
    h->inited = true;
    return LPVS_OK;
}

int32_t lpvs_admm_set_state_f64(lpvs_problem *h, const double *x, const double *z, const double *u, int64_t iters_done) {
    if (!h || !x || !z || !u) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    if (!h->inited) { set_error("lpvs_admm_set_state before lpvs_admm_init"); return LPVS_ESTATE; }
    if (iters_done < 0) { set_error("iters_done must be >= 0"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    LPVS_TRY(copy_state_in(h, h->x.p, x));
    LPVS_TRY(copy_state_in(h, h->z.p, z));
    LPVS_TRY(copy_state_in(h, h->u.p, u));
    const AdmmParams p = make_params(h);
    LPVS_TRY(launch_admm_restate(p, iters_done, s));
    h->fi_sync = -1;                                  // (the next run rebuilds the one-launch iteration's records from the new state)
    h->k_enq = iters_done;
    if (h->xcorr() && iters_done == 0) {              // a restart from iteration 0: the offset vector of a fresh lpvs_admm_init (no correction, no nibble term yet),
        const size_t v = sizeof(double) * (size_t)h->np * (size_t)h->ns;   // not the one the previous run's last scheduled iteration left behind
        LPVS_HIP(hipMemcpyAsync(h->xb.p, h->xb0.p, v, hipMemcpyDeviceToDevice, s));
        if (h->nib_period > 0 && h->xb_corr.p) {
            LPVS_HIP(hipMemcpyAsync(h->xb_corr.p, h->xb0.p, v, hipMemcpyDeviceToDevice, s));
            LPVS_HIP(hipMemsetAsync(h->nib_acc.p, 0, v, s));
        }
    }
    if (h->xcorr() && iters_done > 0)                 // re-entry: the correction of the state handed in (an uninterrupted run holds the one of its last scheduled iteration: same to second order)
    {
        LPVS_TRY(launch_xupdate_correction(make_params(h), h->G.as<double>(), h->M_shift, h->xb_refined ? nullptr : h->bs.as<double>(), h->xb0.as<double>(), h->xb.as<double>(), h->corr.as<double>(), s));
        if (h->nib_period > 0) LPVS_TRY(launch_nibble_refresh(make_params(h), false, nullptr, s, /*split=*/true));
    }
    LPVS_HIP(hipStreamSynchronize(s));
    return LPVS_OK;
}

int32_t lpvs_admm_run(lpvs_problem *h, int64_t max_iters, int64_t *iters_done, double *nxz, int32_t *converged) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    if (!h->inited) { set_error("lpvs_admm_run before lpvs_admm_init"); return LPVS_ESTATE; }
    LPVS_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    AdmmParams p = make_params(h);
    const size_t ns = (size_t)h->ns;
    std::vector<AdmmStatus> st0(ns), st(ns);
    LPVS_HIP(hipMemcpyAsync(st0.data(), h->status.p, sizeof(AdmmStatus) * ns, hipMemcpyDeviceToHost, s));
    LPVS_HIP(hipStreamSynchronize(s));
    bool all0 = true;
    for (auto &q : st0) all0 = all0 && q.converged;
    p.fi_base = st0[0].iters;
    size_t nxc_done = 0;
    if (max_iters > 0 && !all0) {
        if (fi_applicable(p) && h->fi_sync != p.fi_base) LPVS_TRY(launch_fi_setup(p, p.fi_base, false, s));   // state set from outside, or the last chunk took the other path
        LPVS_HIP(hipEventRecord(h->ev[0].a, s));
        int64_t todo = max_iters;
        // (the one-launch iteration of small problems pays two extra launches per chunk: longer chunks)
        const int64_t kGraphIters = small_iter_applicable(p) ? 250 : 50;
        if (h->admm_graph && h->admm_graph_iters != kGraphIters) h->drop_graph();
        if (h->np < kSymmetricMinNp && todo >= 2 * kGraphIters && getenv("LPVS_NO_GRAPH") == nullptr) {
            // two launches of a few microseconds per iteration: host launch cost dominates, so replay a captured
            // chunk.  Iterations past convergence are no-ops (device flag), and exactly max_iters are enqueued.
            if (!h->admm_graph) {
                hipGraph_t g = nullptr;
                LPVS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                const int32_t rc = launch_admm_iterations(p, kGraphIters, s);
                const hipError_t e = hipStreamEndCapture(s, &g);
                if (rc != LPVS_OK || e != hipSuccess || g == nullptr) {
                    (void)hipGetLastError();
                    if (g) (void)hipGraphDestroy(g);
                    set_error("hipGraph capture of the ADMM chunk failed");
                    return LPVS_EDEVICE;
                }
                const hipError_t ei = hipGraphInstantiate(&h->admm_graph, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (ei != hipSuccess) { (void)hipGetLastError(); h->admm_graph = nullptr; set_error("hipGraphInstantiate failed"); return LPVS_EDEVICE; }
                h->admm_graph_iters = kGraphIters;
            }
            while (todo >= h->admm_graph_iters) { LPVS_HIP(hipGraphLaunch(h->admm_graph, s)); todo -= h->admm_graph_iters; }
        }
        size_t nxc = 0;                              // corrections of this call (event pairs xc_ev[2 i], xc_ev[2 i + 1])
        while (todo > 0) {
            // the correction schedule cuts the run at the iterations base^j, whatever chunks the caller asks for: the iterates do not depend on the chunking
            int64_t step = todo;
            long long next_corr = 0;
            if (h->xcorr()) {
                next_corr = h->next_correction(h->k_enq);
                if (next_corr - h->k_enq < step) step = next_corr - h->k_enq;
            }
            LPVS_TRY(launch_admm_iterations(p, step, s));
            if (p.nib_period > 0) for (long long g = p.fi_base; g < p.fi_base + step; ++g) h->n_nib += nib_refresh_due(g, p.nib_period, p.nib_ramp);
            todo -= step; h->k_enq += step;
            if (h->xcorr() && h->k_enq == next_corr) {
                if (h->xc_ev.size() < 2 * (nxc + 1)) {
                    hipEvent_t ea = nullptr, eb = nullptr;
                    LPVS_HIP(hipEventCreate(&ea)); h->xc_ev.push_back(ea);
                    LPVS_HIP(hipEventCreate(&eb)); h->xc_ev.push_back(eb);
                }
                LPVS_HIP(hipEventRecord(h->xc_ev[2 * nxc], s));
                LPVS_TRY(launch_xupdate_correction(p, h->G.as<double>(), h->M_shift, h->xb_refined ? nullptr : h->bs.as<double>(), h->xb0.as<double>(), h->xb.as<double>(), h->corr.as<double>(), s));
                if (p.nib_period > 0) LPVS_TRY(launch_nibble_refresh(p, false, nullptr, s, /*split=*/true));   // xb_corr = xb - N rhs: the refreshes re-add the nibble term of THEIR right-hand side
                LPVS_HIP(hipEventRecord(h->xc_ev[2 * nxc + 1], s));
                ++nxc;
                if (todo > 0 && !(h->tol > 0)) p.fi_base += step;   // (no stopping test can fire: the device committed exactly `step` more)
                else if (todo > 0) {   // the next sub-chunk starts from the iteration count the device committed (a stopping test may have fired)
                    LPVS_HIP(hipMemcpyAsync(st.data(), h->status.p, sizeof(AdmmStatus) * ns, hipMemcpyDeviceToHost, s));
                    LPVS_HIP(hipStreamSynchronize(s));
                    bool alls = true;
                    for (auto &q : st) alls = alls && q.converged;
                    if (alls) break;
                    p.fi_base = st[0].iters;
                }
            }
        }
        LPVS_HIP(hipEventRecord(h->ev[0].b, s));
        nxc_done = nxc;
    }
    LPVS_HIP(hipMemcpyAsync(st.data(), h->status.p, sizeof(AdmmStatus) * ns, hipMemcpyDeviceToHost, s));
    LPVS_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < nxc_done; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->xc_ev[2 * i], h->xc_ev[2 * i + 1]) == hipSuccess) { h->t_xcorr += ms; h->n_xcorr += 1; } else (void)hipGetLastError();
    }
    long long it_max = 0, it0_max = 0; double nxz_max = 0; bool all = true;
    for (size_t q = 0; q < ns; ++q) {
        if (st[q].iters > it_max) it_max = st[q].iters;
        if (st0[q].iters > it0_max) it0_max = st0[q].iters;
        if (st[q].nxz > nxz_max) nxz_max = st[q].nxz;
        all = all && st[q].converged;
    }
    if (max_iters > 0 && !all0) { h->t_admm += h->ev[0].ms(); h->admm_iters_timed += (double)(it_max - it0_max); }
    if (max_iters > 0 && !all0) h->fi_sync = fi_applicable(p) ? (long long)st[0].iters : -1;   // (the other path leaves no records behind)
    if (iters_done) *iters_done = it_max;     // ns > 1: the slowest signal; per-signal values via lpvs_admm_status
    if (nxz) *nxz = nxz_max;
    if (converged) *converged = all ? 1 : 0;
    return LPVS_OK;
}

/* 0: full symmetric matrix (plain mat-vec, n < 2048), 1: tile-packed doubles, 2: tile-packed floats, 3: tile-packed split, 4: mixed;
   + 16: the iteration runs as ONE launch (fixed-point accumulation of the tile partials, update in the next launch's prologue);
   + 32: the mixed storage's fixed-point tiles keep 32 significant bits (4 B per element) */
int32_t lpvs_admm_matvec_kind(lpvs_problem *h, int32_t *kind) {
    if (!h || !kind) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    if (!h->inited) { set_error("lpvs_admm_matvec_kind before lpvs_admm_init"); return LPVS_ESTATE; }
    *kind = h->np >= kSymmetricMinNp ? h->Mp_mode : kMpNone;
    if (fi_applicable(make_params(h)) || small_iter_applicable(make_params(h))) *kind |= 16;   // one launch per iteration (with the prox currently set)
    if (make_params(h).mp_fix32) *kind |= 32;                                                  // 32-bit fixed-point tiles
    return LPVS_OK;
}

int32_t lpvs_admm_time_matvec(lpvs_problem *h, int32_t reps, double *us_per_launch, double *bytes_per_launch) {
    if (!h || !us_per_launch) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    if (!h->inited) { set_error("lpvs_admm_time_matvec before lpvs_admm_init"); return LPVS_ESTATE; }
    if (reps <= 0) { set_error("reps must be positive"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const bool sym = h->np >= kSymmetricMinNp;
    const AdmmParams p = make_params(h);
    LPVS_TRY(launch_admm_matvec_only(p, 3, s));   // warm
    LPVS_HIP(hipEventRecord(h->ev[1].a, s));
    LPVS_TRY(launch_admm_matvec_only(p, reps, s));
    LPVS_HIP(hipEventRecord(h->ev[1].b, s));
    LPVS_HIP(hipStreamSynchronize(s));
    *us_per_launch = h->ev[1].ms() * 1e3 / reps;
    h->nib_us = 0;
    if (p.nib_period > 0 && !nib_fused_applies(p)) {   // one refresh of the stale nibble product where it is three kernels of its own (the two-launch iteration), stand-alone; the offset vector is put back
        const size_t v = sizeof(double) * (size_t)h->np;
        LPVS_HIP(hipMemcpyAsync(h->nib_rhs.p, h->xb.p, v, hipMemcpyDeviceToDevice, s));
        LPVS_TRY(launch_nibble_refresh(p, false, nullptr, s));
        LPVS_HIP(hipEventRecord(h->ev[1].a, s));
        for (int i = 0; i < reps; ++i) LPVS_TRY(launch_nibble_refresh(p, false, nullptr, s));
        LPVS_HIP(hipEventRecord(h->ev[1].b, s));
        LPVS_HIP(hipMemcpyAsync(h->xb.p, h->nib_rhs.p, v, hipMemcpyDeviceToDevice, s));
        LPVS_HIP(hipStreamSynchronize(s));
        h->nib_us = h->ev[1].ms() * 1e3 / reps;
    }
    if (bytes_per_launch) *bytes_per_launch = sym ? h->Mp_stream_bytes - (make_params(h).mp_fix32 ? 8192.0 * (double)h->Mp_fixed_tiles : 0.0) : 8.0 * (double)h->np * (double)h->np;   // (32-bit reads: no nibble plane)
    return LPVS_OK;
}

int32_t lpvs_problem_num_signals(const lpvs_problem *h, int64_t *ns) {
    if (!h || !ns) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    *ns = h->ns;
    return LPVS_OK;
}

int32_t lpvs_admm_status(lpvs_problem *h, int64_t signal, int64_t *iters_done, double *nxz, int32_t *converged) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    if (!h->inited) { set_error("lpvs_admm_status before lpvs_admm_init"); return LPVS_ESTATE; }
    if (signal < 0 || signal >= h->ns) { set_error("signal %lld out of range", (long long)signal); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    AdmmStatus st{};
    LPVS_HIP(hipMemcpyAsync(&st, h->status.as<AdmmStatus>() + signal, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    LPVS_HIP(hipStreamSynchronize(h->stream));
    if (iters_done) *iters_done = st.iters;
    if (nxz) *nxz = st.nxz;
    if (converged) *converged = st.converged;
    return LPVS_OK;
}

// the offset vector(s): xb, and -- handles that run the stale nibble product -- xb_corr behind it (both are state between two refreshes)
static bool offset_has_nibble_part(const lpvs_problem *h) { return h->nib_period > 0 && h->xb_corr.p != nullptr; }
int32_t lpvs_admm_offset_len(lpvs_problem *h, int64_t *len) {
    if (!h || !len) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    if (!h->inited) { set_error("lpvs_admm_offset_len before lpvs_admm_init"); return LPVS_ESTATE; }
    *len = !h->offset_form || !h->xb.p ? 0 : (int64_t)h->ns * h->n * (offset_has_nibble_part(h) ? 2 : 1);
    return LPVS_OK;
}
static int32_t offset_len_matches(const lpvs_problem *h, int64_t len) {
    const int64_t want = (offset_has_nibble_part(h) ? 2 : 1) * (int64_t)h->ns * h->n;
    if (len != want) {
        set_error("offset buffer of %lld doubles, this handle's offset vector has %lld (lpvs_admm_offset_len: n x ns, or 2 n for a handle that iterates on "
                  "LPVS_STORAGE_MIXED32 reads -- a checkpoint goes back into a handle with the same storage option)", (long long)len, (long long)want);
        return LPVS_EARGUMENT;
    }
    return LPVS_OK;
}
int32_t lpvs_admm_get_offset_f64(lpvs_problem *h, double *xb_out, int64_t len) {
    if (!h || !xb_out) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    if (!h->inited || !h->offset_form || !h->xb.p) { set_error("this handle has no offset vector (n < 2048, or lpvs_admm_init has not run)"); return LPVS_ESTATE; }
    LPVS_TRY(offset_len_matches(h, len));
    LPVS_HIP(hipSetDevice(h->device));
    LPVS_TRY(copy_state_out(h, h->xb.p, xb_out));
    if (offset_has_nibble_part(h)) LPVS_TRY(copy_state_out(h, h->xb_corr.p, xb_out + (int64_t)h->ns * h->n));
    return LPVS_OK;
}
int32_t lpvs_admm_set_offset_f64(lpvs_problem *h, const double *xb, int64_t len) {
    if (!h || !xb) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    if (!h->inited || !h->offset_form || !h->xb.p) { set_error("this handle has no offset vector (n < 2048, or lpvs_admm_init has not run)"); return LPVS_ESTATE; }
    LPVS_TRY(offset_len_matches(h, len));           // (no length, no way to tell a 36-bit handle's n-vector from a 32-bit handle's 2n: a host out-of-bounds read)
    LPVS_HIP(hipSetDevice(h->device));
    LPVS_TRY(copy_state_in(h, h->xb.p, xb));
    if (offset_has_nibble_part(h)) LPVS_TRY(copy_state_in(h, h->xb_corr.p, xb + (int64_t)h->ns * h->n));
    return LPVS_OK;
}

int32_t lpvs_admm_get_f64(lpvs_problem *h, double *x_out, double *z_out, double *u_out) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    if (!h->inited) { set_error("lpvs_admm_get before lpvs_admm_init"); return LPVS_ESTATE; }
    LPVS_HIP(hipSetDevice(h->device));
    if (x_out) LPVS_TRY(copy_state_out(h, h->x.p, x_out));
    if (z_out) LPVS_TRY(copy_state_out(h, h->z.p, z_out));
    if (u_out) LPVS_TRY(copy_state_out(h, h->u.p, u_out));
    return LPVS_OK;
}

static int32_t pack_host(const lpvs_problem *h, const double *c, double *re_out, double *im_out) {
    const int64_t Nf = h->Nf, nb = h->nb;
    const int64_t m = h->kind == 1 ? Nf * nb : Nf;
    std::vector<double> re((size_t)m), im((size_t)m);
    if (h->kind == 0) {  // fourier2complex, src/utilities.jl:62-73
        if (!h->zerofreq) for (int64_t i = 0; i < Nf; ++i) { re[i] = c[i]; im[i] = c[Nf + i]; }
        else { re[0] = c[0]; im[0] = 0.0; for (int64_t i = 1; i < Nf; ++i) { re[i] = c[i]; im[i] = c[Nf + i - 1]; } }
    } else if (h->kind == 1) {  // z[sortperm(inds)] -> complex, src/lasso.jl:67-68
        for (int64_t f = 0; f < Nf; ++f)
            for (int64_t v = 0; v < nb; ++v) { re[f + v * Nf] = c[f * 2 * nb + v]; im[f + v * Nf] = c[f * 2 * nb + nb + v]; }
    } else { set_error("explicit-Gram problems have no parameter packing"); return LPVS_EUNSUPPORTED; }
    const size_t bytes = sizeof(double) * (size_t)m;
    if (re_out) { if (is_device_ptr(re_out)) { LPVS_HIP(hipMemcpy(re_out, re.data(), bytes, hipMemcpyHostToDevice)); } else memcpy(re_out, re.data(), bytes); }
    if (im_out) { if (is_device_ptr(im_out)) { LPVS_HIP(hipMemcpy(im_out, im.data(), bytes, hipMemcpyHostToDevice)); } else memcpy(im_out, im.data(), bytes); }
    return LPVS_OK;
}

int32_t lpvs_problem_get_params_f64(lpvs_problem *h, int32_t which, double *re_out, double *im_out) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    if (!h->inited) { set_error("lpvs_problem_get_params before lpvs_admm_init"); return LPVS_ESTATE; }
    LPVS_HIP(hipSetDevice(h->device));
    std::vector<double> c((size_t)h->n * (size_t)h->ns);
    LPVS_TRY(copy_state_out(h, which == 1 ? h->x.p : h->z.p, c.data()));
    const int64_t m = h->kind == 1 ? h->Nf * h->nb : h->Nf;
    for (int64_t q = 0; q < h->ns; ++q)   // signal q -> columns q of the m x ns outputs
        LPVS_TRY(pack_host(h, c.data() + q * h->n, re_out ? re_out + q * m : nullptr, im_out ? im_out + q * m : nullptr));
    return LPVS_OK;
}

int32_t lpvs_problem_pack_params_f64(lpvs_problem *h, const double *coef, double *re_out, double *im_out) {
    if (!h || !coef) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    std::vector<double> c;
    LPVS_TRY(fetch_host(c, coef, h->n));
    return pack_host(h, c.data(), re_out, im_out);
}

int32_t lpvs_problem_get_timing(lpvs_problem *h, double *out, int32_t n_out) {
    if (!h || !out) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    const double v[13] = {h->t_basis, h->t_gram, h->t_reduce, h->t_factor, h->t_admm, h->gram_launches, h->gram_flops, h->admm_iters_timed, h->gram_form,
                          h->t_xcorr, h->n_xcorr, h->n_nib, h->nib_us};
    for (int i = 0; i < n_out && i < 13; ++i) out[i] = v[i];
    return LPVS_OK;
}

// development aid: LPVS_TRACE=1 prints host wall-clock per phase of the batched path (with stream syncs)
struct PhaseTrace {
    bool on; hipStream_t s; double t0;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
    PhaseTrace(hipStream_t s_) : on(getenv("LPVS_TRACE") != nullptr), s(s_), t0(now()) {}
    void mark(const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const double t = now();
        fprintf(stderr, "[lpvs trace] %-28s %9.3f ms\n", what, (t - t0) * 1e3);
        t0 = t;
    }
};

// ---- ls_spectral: primal or dual normal equations ---------------------------------------------------------
int32_t lpvs_ls_spectral_f64(const double *y, const double *t, int64_t N, const double *f, int64_t Nf, double lam, int32_t device,
                             double *re_out, double *im_out) {
    if (N <= 0 || Nf <= 0) { set_error("N and Nf must be positive"); return LPVS_EARGUMENT; }
    int64_t zf = 0;
    LPVS_TRY(lpvs_check_freq_f64(f, Nf, &zf));
    const int64_t nreg = zf ? 2 * Nf - 1 : 2 * Nf;
    lpvs_problem *h = nullptr;
    if (N >= nreg) {   // tall: the ordinary Gram problem
        LPVS_TRY(lpvs_problem_create_fourier_f64(y, t, N, f, Nf, nullptr, device, &h));
        std::vector<double> x((size_t)nreg);
        int32_t rc = lpvs_problem_solve_ridge_f64(h, lam * lam, x.data());
        if (rc == LPVS_OK) rc = lpvs_problem_pack_params_f64(h, x.data(), re_out, im_out);
        lpvs_problem_destroy(h);
        return rc;
    }
    // fat: Gram over the regressor columns, G2 = A A' (N x N), alpha = (G2 + lam^2 I)^-1 y, x = A' alpha
    lpvs_problem *dummy = nullptr;
    LPVS_TRY(problem_begin(device, &dummy, &h));
    struct Guard { lpvs_problem *h; ~Guard() { delete h; } } guard{h};
    hipStream_t s = h->stream;
    h->kind = 0; h->N = nreg; h->Nf = Nf; h->zerofreq = zf; h->n = N; h->np = round_up(N, 128);
    const GramPlan pl = make_gram_plan(N, nreg);
    const int64_t rows = pl.ksplit * pl.rows_per_chunk, ldn = round_up(N, 256);
    DevArg dy, dt, df;
    LPVS_TRY(dy.set(y, N, s)); LPVS_TRY(dt.set(t, N, s)); LPVS_TRY(df.set(f, Nf, s));
    LPVS_TRY(h->G.alloc(sizeof(double) * (size_t)h->np * (size_t)h->np));
    LPVS_TRY(h->b.alloc(sizeof(double) * (size_t)h->np));
    LPVS_HIP(hipMemsetAsync(h->G.p, 0, h->G.bytes, s));
    LPVS_HIP(hipMemsetAsync(h->b.p, 0, h->b.bytes, s));
    LPVS_TRY(alloc_state(h));
    LPVS_HIP(hipMemcpyAsync(h->b.p, dy.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, s));   // right-hand side = y
    DevBuf D, slab, xo;
    DrainOnExit drain(s);
    LPVS_TRY(D.alloc(sizeof(double) * (size_t)rows * (size_t)ldn));
    LPVS_TRY(launch_fourier_dual_panel(dt.p, N, df.p, Nf, (int)zf, D.as<double>(), ldn, rows, s));
    LPVS_TRY(slab.alloc(pl.slab_bytes));
    LPVS_TRY(launch_gram_panel(pl, D.as<double>(), ldn, nullptr, slab.as<double>(), s));
    LPVS_TRY(launch_gram_reduce(pl, slab.as<double>(), h->G.as<double>(), h->np, s));
    LPVS_TRY(factorize(h, lam * lam));
    LPVS_TRY(launch_symv(h->M.as<double>(), h->np, h->b.as<double>(), h->scratch.as<double>(), s));       // alpha (pad = 0)
    LPVS_TRY(xo.alloc(sizeof(double) * (size_t)rows));
    // x[c] = sum_n D[c][n] alpha[n]; alpha is zero beyond N and np <= ldn may not hold, so use min(np, ldn) columns
    const int64_t cols = h->np < ldn ? h->np : ldn;
    LPVS_TRY(launch_rect_matvec(D.as<double>(), nreg, cols, ldn, h->scratch.as<double>(), xo.as<double>(), s));
    std::vector<double> x((size_t)nreg);
    LPVS_TRY(copy_from_device(x.data(), xo.p, sizeof(double) * (size_t)nreg, s));
    // pack with the Fourier layout (kind 0, Nf, zerofreq are set; n is the dual size, so pack by hand)
    std::vector<double> re((size_t)Nf), im((size_t)Nf);
    if (!zf) for (int64_t i = 0; i < Nf; ++i) { re[i] = x[i]; im[i] = x[Nf + i]; }
    else { re[0] = x[0]; im[0] = 0.0; for (int64_t i = 1; i < Nf; ++i) { re[i] = x[i]; im[i] = x[Nf + i - 1]; } }
    const size_t bytes = sizeof(double) * (size_t)Nf;
    if (re_out) { if (is_device_ptr(re_out)) { LPVS_HIP(hipMemcpy(re_out, re.data(), bytes, hipMemcpyHostToDevice)); } else memcpy(re_out, re.data(), bytes); }
    if (im_out) { if (is_device_ptr(im_out)) { LPVS_HIP(hipMemcpy(im_out, im.data(), bytes, hipMemcpyHostToDevice)); } else memcpy(im_out, im.data(), bytes); }
    LPVS_HIP(hipStreamSynchronize(s));
    return LPVS_OK;
}

// ---- batched windows ------------------------------------------------------------------------------------
// One engine behind ls_windowpsd / ls_windowcsd / ls_cohere (src/lsfft.jl:112-193): all windows [win_lo, win_hi) of
// Windows2 / Windows3 (src/windows.jl:27-36, :94-103) are solved together; `ns` signals sharing t (the y and u of the
// cross-spectral drivers) share every window's Gram and factorisation and differ only in the right-hand side.
}  // extern "C"

namespace {

// phase times (ms) of the calling thread's last engine call: [0] Gram + rhs, [1] inverse (+ pack), [2] ADMM iterations or dense
// solves, [3] windows, [4] batch mat-vec microseconds per launch (LPVS_WINDOW_MATVEC_TIMING=1 only, last pass), [5] windows of
// that pass, [6] passes, [7] 1 if the structured Gram was used
thread_local double g_win_multi[2] = {0, 0};   // last several-device call of this thread: ranks of the RCCL communicator that gathered (0: no collective), devices driven
thread_local double g_win_timing[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [8] = bytes of packed inverses one timed mat-vec launch reads

// sink(window index relative to win_lo, signal, re[Nf], im[Nf], iterations) is called in window order, signals innermost
template <class Sink>
int32_t windows_engine(const WinJob &a, Sink sink) {
    int64_t k = 0;
    LPVS_TRY(lpvs_window_count(a.L, a.n, a.noverlap, &k));
    const int64_t n = a.n, Nf = a.Nf, ns = a.ns;
    const int64_t noverlap = a.noverlap < 0 ? n >> 1 : a.noverlap;
    const int64_t win_lo = a.win_lo, win_hi = a.win_hi;
    const bool sparse = a.estimator == LPVS_EST_SPARSE || a.estimator == LPVS_EST_SPARSE_INIT;
    const bool init = a.estimator == LPVS_EST_SPARSE_INIT;       // x0 = fourier_solve(A, y, zerofreq, lam), src/lasso.jl:112
    if (!sparse && a.estimator != LPVS_EST_DENSE) { set_error("unknown estimator %d", a.estimator); return LPVS_EARGUMENT; }
    if (ns < 1 || ns > 64) { set_error("number of signals %lld outside [1, 64]", (long long)ns); return LPVS_EARGUMENT; }
    if (win_lo < 0 || win_hi > k || win_lo > win_hi) { set_error("window range [%lld,%lld) outside [0,%lld)", (long long)win_lo, (long long)win_hi, (long long)k); return LPVS_EARGUMENT; }
    const double mu = a.mu, tol = a.tol;
    if (sparse) {
        if (!(mu >= 0 && mu <= 1)) { set_error("μ should be ≤ 1"); return LPVS_EASSERT; }
        if (mu == 0) { set_error("mu = 0 makes the x-update singular"); return LPVS_ENUMERIC; }
        if (a.prox_kind < LPVS_PROX_L1 || a.prox_kind > LPVS_PROX_GROUP_L2) { set_error("unknown prox kind %d", a.prox_kind); return LPVS_EUNSUPPORTED; }
        if (a.prox_kind == LPVS_PROX_GROUP_L2 && a.group_len <= 0) { set_error("group_len must be positive"); return LPVS_EARGUMENT; }
        if (a.prox_kind == LPVS_PROX_GROUP_L2 && a.group_len > 8192) { set_error("group_len > 8192 is not supported by the device prox"); return LPVS_EUNSUPPORTED; }
        if (a.linear_sign != 1 && a.linear_sign != -1) { set_error("linear_sign must be +1 or -1"); return LPVS_EARGUMENT; }
    }
    int64_t zf = 0;
    LPVS_TRY(lpvs_check_freq_f64(a.freqs, Nf, &zf));
    for (double &v : g_win_timing) v = 0;
    g_win_multi[0] = g_win_multi[1] = 0;
    const int64_t nwin = win_hi - win_lo;
    if (nwin == 0) return LPVS_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); set_error("no HIP device visible (the gfx950 path has no CPU fallback)"); return LPVS_EDEVICE; }
    if (a.device < 0 || a.device >= count) { set_error("device %d out of range", a.device); return LPVS_EDEVICE; }
    LPVS_HIP(hipSetDevice(a.device));
    StreamBundle *res = bundle_acquire(a.device);
    if (!res) { set_error("stream / event creation failed"); return LPVS_EDEVICE; }
    struct BundleGuard { StreamBundle *b; ~BundleGuard() { bundle_release(b); } } bg{res};
    hipStream_t s = res->stream;
    EventPair *ev = res->ev;

    const int64_t nreg = zf ? 2 * Nf - 1 : 2 * Nf, np = round_up(nreg, 128), ld = round_up(nreg, 256);
    // sub-batch: the k-major regressor panels of one pass stay under a budget (default 48 GiB of the 288 GB)
    size_t budget = (size_t)48 << 30;
    if (const char *e = getenv("LPVS_BATCH_PANEL_GIB")) { const long g = atol(e); if (g > 0) budget = (size_t)g << 30; }
    int64_t bw = (int64_t)(budget / (sizeof(double) * (size_t)round_up(n, 64) * (size_t)ld));
    if (bw < 1) bw = 1;
    if (bw > nwin) bw = nwin;
    if (bw > 8192) bw = 8192;
    const int64_t step = n - noverlap;
    std::vector<DevArg> dys((size_t)ns);
    DevArg dt, df;
    for (int64_t q = 0; q < ns; ++q) LPVS_TRY(dys[(size_t)q].set(a.ys[q], a.L, s));
    LPVS_TRY(dt.set(a.t, a.L, s)); LPVS_TRY(df.set(a.freqs, Nf, s));
    // structured Gram (nudft.hip) when T(2pi)*freqs is an arithmetic progression: no regressor panels at all
    ApSlots sl;
    int jopt[kOptCount];                             // the caller's options: captured by the entry point, or this thread's own
    if (a.opt_captured) for (int i = 0; i < kOptCount; ++i) jopt[i] = a.opt[i];
    else capture_default_options(jopt);
    {
        const int form_opt = option_in_effect(LPVS_OPT_GRAM_FORM, jopt[LPVS_OPT_GRAM_FORM]);
        const std::string form = form_opt == LPVS_GRAM_AP ? "ap" : form_opt == LPVS_GRAM_KRS ? "krs" : form_opt == LPVS_GRAM_KR ? "kr" : "auto";
        if (form == "auto" || form == "ap") {
            std::vector<double> hw;
            LPVS_TRY(fetch_host(hw, a.freqs, Nf));
            for (auto &v : hw) v = 6.283185307179586 * v;
            double tlo, thi, tam = a.t_absmax;
            if (!(tam >= 0)) LPVS_TRY(device_minmax(dt.p, a.L, &tlo, &thi, &tam, s));
            sl = make_ap_slots(hw, tam);
            if (form == "ap" && !sl.ok) { set_error("LPVS_GRAM_FORM=ap but 2*pi*freqs is not an arithmetic progression (max|eps|*max|t| = %.3g)", sl.emax * tam); return LPVS_EARGUMENT; }
        }
    }
    const bool ap = sl.ok;
    const int nmat = (sparse ? 2 : 3) + (init ? 2 : 0);   // resident np x np matrices per window: M, packed M (sparse) / Q, M, work (dense) (+ A'A and its inverse for init)
    if (ap) {   // windows per pass bounded by the matrices: 32 GiB
        bw = (int64_t)(((size_t)32 << 30) / (sizeof(double) * (size_t)np * (size_t)np * (size_t)nmat));
        if (bw < 1) bw = 1;
        if (bw > nwin) bw = nwin;
        if (bw > 8192) bw = 8192;
    }
    // the sample split of the dense form is chosen for a nominal batch of 64 windows, whatever the shard holds: the
    // summation order of a window must not depend on how the windows are sharded
    const GramPlan pl = make_gram_plan(nreg, n, 64);
    const int64_t nrows = ap ? n : pl.ksplit * pl.rows_per_chunk;
    const size_t panel_bytes = ap ? 0 : sizeof(double) * (size_t)nrows * (size_t)ld;
    while (!ap && bw > 1 && panel_bytes * (size_t)bw > budget) --bw;
    // structured form: every window is cut into segments of rpcw samples, one workgroup row per segment; the segment length
    // is fixed: the summation order of a window must not depend on how many windows share the pass, so that window shards
    // (ranks of a node) reproduce the whole run bit for bit
    const int64_t rpcw = ap ? std::min<int64_t>(n, 4096) : 0;
    const int spw = ap ? (int)ceil_div(n, rpcw) : 0;
    DevBuf Wp;
    const double *Wdev = nullptr;
    if (a.W != nullptr) {
        LPVS_TRY(Wp.alloc(sizeof(double) * (size_t)nrows));
        LPVS_HIP(hipMemsetAsync(Wp.p, 0, Wp.bytes, s));
        LPVS_TRY(copy_to_device(Wp.p, a.W, sizeof(double) * (size_t)n, s));
        Wdev = Wp.as<double>();
    }
    PhaseTrace tr(s);
    DevBuf P, slab, M, Q, bvec, x, z, u, rhs, xb, status, work, istat, offs, scr, part, Mp, seg, npart, tab, tabb, fibuf, Q0, M0, b0, x0buf, ballscr, xbc, nibacc;
    ApSlotsDev sd;
    DrainOnExit drain(s);
    const int64_t nprob_max = bw * ns;
    // slot sums by non-uniform FFT (nufft.hip) when one progression serves all slots; LPVS_NUDFT=direct keeps the direct sums
    const bool wnufft = ap && sl.merged && nufft_windows_applicable(n, sl.nsl) &&
                        option_in_effect(LPVS_OPT_SLOT_SUMS, jopt[LPVS_OPT_SLOT_SUMS]) != LPVS_SLOTS_DIRECT;
    const bool wnufft_rhs = wnufft && sl.s0 % 2 == 0 && sl.s0 / 2 + sl.nf8 <= sl.nsl;   // a + f D = mode s0/2 + f (residual delta/2 -> eps)
    const int nfg = wnufft ? nufft_grid_size(sl.nsl) : 0;
    const int64_t wcols = wnufft_rhs ? 1 + ns : 1;
    DevBuf wgrids, wscale_g, wscale_r, wepsr;
    if (wnufft) {
        LPVS_TRY(wgrids.alloc(sizeof(double) * (size_t)bw * (size_t)wcols * 2 * (size_t)nfg));
        const std::vector<double> sg = nufft_window_scale(nfg, 0, (int)sl.nsl);
        LPVS_TRY(wscale_g.alloc(sizeof(double) * sg.size()));
        LPVS_TRY(copy_to_device(wscale_g.p, sg.data(), sizeof(double) * sg.size(), s));
        if (wnufft_rhs) {
            const std::vector<double> sr = nufft_window_scale(nfg, (int)(sl.s0 / 2), (int)sl.nf8);
            std::vector<double> er(sl.eps.size());
            for (size_t f = 0; f < er.size(); ++f) er[f] = sl.eps[f] + 0.5 * sl.delta;
            LPVS_TRY(wscale_r.alloc(sizeof(double) * sr.size()));
            LPVS_TRY(wepsr.alloc(sizeof(double) * er.size()));
            LPVS_TRY(copy_to_device(wscale_r.p, sr.data(), sizeof(double) * sr.size(), s));
            LPVS_TRY(copy_to_device(wepsr.p, er.data(), sizeof(double) * er.size(), s));
        }
        LPVS_HIP(hipStreamSynchronize(s));           // (the host vectors are locals)
    }
    if (ap) {
        LPVS_TRY(sd.upload(sl, s));
        LPVS_TRY(seg.alloc(sizeof(int64_t) * 3 * (size_t)bw * (size_t)spw));
        LPVS_TRY(npart.alloc(sizeof(double) * (size_t)bw * (size_t)spw * (size_t)sl.nsl * 4));
        LPVS_TRY(tab.alloc(sizeof(double) * (size_t)bw * (size_t)sl.nsl * 4));
        LPVS_TRY(tabb.alloc(sizeof(double) * (size_t)bw * (size_t)sl.nf8 * 4));
    } else {
        LPVS_TRY(P.alloc(panel_bytes * (size_t)bw));
        LPVS_TRY(slab.alloc(pl.slab_bytes * (size_t)bw));
    }
    LPVS_TRY(M.alloc(sizeof(double) * (size_t)np * (size_t)np * (size_t)bw));
    if (sparse) {
        LPVS_TRY(part.alloc(sizeof(double) * symv_part_doubles(np, nprob_max)));
        LPVS_TRY(Mp.alloc(sizeof(double) * symv_packed_doubles(np) * (size_t)bw));
        LPVS_TRY(status.alloc(sizeof(AdmmStatus) * (size_t)nprob_max));
    } else {
        LPVS_TRY(Q.alloc(sizeof(double) * (size_t)np * (size_t)np * (size_t)bw));
    }
    const size_t vb = sizeof(double) * (size_t)np * (size_t)nprob_max;
    LPVS_TRY(bvec.alloc(vb)); LPVS_TRY(x.alloc(vb)); LPVS_TRY(z.alloc(vb)); LPVS_TRY(u.alloc(vb)); LPVS_TRY(rhs.alloc(vb));
    const int storage_opt = option_in_effect(LPVS_OPT_M_STORAGE, jopt[LPVS_OPT_M_STORAGE]);
    const bool split_storage = sparse && storage_opt != LPVS_STORAGE_F64;
    if (split_storage) LPVS_TRY(xb.alloc(vb));
    LPVS_TRY(istat.alloc(sizeof(int) * (size_t)bw));
    LPVS_TRY(work.alloc(spd_inverse_work_bytes(np) * (size_t)bw));
    LPVS_TRY(offs.alloc(sizeof(int64_t) * (size_t)bw));
    if (!ap) LPVS_TRY(scr.alloc(rhs_scratch_bytes(n, nreg) * (size_t)bw));
    if (init) {
        LPVS_TRY(Q0.alloc(sizeof(double) * (size_t)np * (size_t)np * (size_t)bw)); LPVS_TRY(M0.alloc(sizeof(double) * (size_t)np * (size_t)np * (size_t)bw));
        LPVS_TRY(b0.alloc(vb)); LPVS_TRY(x0buf.alloc(vb));
    }
    if (sparse && a.prox_kind == LPVS_PROX_BALL_L0 && nreg > 8192) LPVS_TRY(ballscr.alloc(2 * vb));   // selection keys of vectors that do not fit the LDS image

    tr.mark("alloc");
    std::vector<int64_t> hseg((size_t)3 * (size_t)bw * (size_t)spw);
    std::vector<double> zh((size_t)np * (size_t)nprob_max), re((size_t)Nf), im((size_t)Nf);
    std::vector<int64_t> hoff((size_t)bw);
    std::vector<AdmmStatus> hst((size_t)nprob_max);
    std::vector<int> hist_((size_t)bw);
    const bool want_mv = sparse && getenv("LPVS_WINDOW_MATVEC_TIMING") != nullptr;
    g_win_timing[3] = (double)nwin; g_win_timing[7] = ap ? (wnufft ? 2 : 1) : 0;   // Gram form: 0 dense, 1 structured (direct sums), 2 structured (NUFFT)

    for (int64_t w0 = 0; w0 < nwin; w0 += bw) {
        const int nb_ = (int)((nwin - w0 < bw) ? nwin - w0 : bw);
        const int nprob = nb_ * (int)ns;
        for (int q = 0; q < nb_; ++q) hoff[q] = (win_lo + w0 + q) * step;           // arraysplit offsets, src/windows.jl:33
        LPVS_HIP(hipMemcpyAsync(offs.p, hoff.data(), sizeof(int64_t) * (size_t)nb_, hipMemcpyHostToDevice, s));
        double *G = sparse ? M.as<double>() : Q.as<double>();                      // where the window Grams are assembled
        LPVS_HIP(hipEventRecord(ev[0].a, s));
        // Gram A' diag(Wd) A of every window of the pass -> Gdst ([nb_][np][np]) and right-hand sides A' diag(Wd) y_s -> bdst ([nprob][np])
        auto build_gram = [&](const double *Wd, double *Gdst, double *bdst) -> int32_t {
            if (ap) {
                for (int q = 0; q < nb_; ++q)
                    for (int c = 0; c < spw; ++c) {
                        int64_t *e = hseg.data() + 3 * ((size_t)q * (size_t)spw + (size_t)c);
                        e[0] = hoff[q] + (int64_t)c * rpcw;
                        e[1] = std::min(hoff[q] + n, e[0] + rpcw);
                        e[2] = hoff[q];
                    }
                LPVS_TRY(copy_to_device(seg.p, hseg.data(), sizeof(int64_t) * 3 * (size_t)nb_ * (size_t)spw, s));
                LPVS_HIP(hipMemsetAsync(Gdst, 0, sizeof(double) * (size_t)np * (size_t)np * (size_t)nb_, s));
                LPVS_HIP(hipMemsetAsync(bdst, 0, vb, s));
                const int64_t gstride = wcols * 2 * (int64_t)nfg;      // doubles per window in wgrids: [column][plain, x-weighted][nfg]
                if (wnufft) {   // column 0 = the window weights alone (Gram); with it, in the same pass over the samples, signal 0
                    LPVS_TRY(launch_nufft_window_spread(dt.p, Wd, nullptr, wnufft_rhs ? dys[0].p : nullptr, wnufft_rhs, offs.as<int64_t>(), nb_, n, sl.step.hi[1],
                                                        sl.step.lo[1], nfg, wgrids.as<double>(), wgrids.as<double>() + 2 * (int64_t)nfg, gstride, s));
                    LPVS_TRY(launch_nufft_window_modes(wgrids.as<double>(), gstride, nb_, nfg, 0, (int)sl.nsl, wscale_g.as<double>(), tab.as<double>(), sl.nsl * 4, s));
                } else {
                    LPVS_TRY(launch_nudft_windows(dt.p, nullptr, Wd, sd.hi.as<double>(), sd.lo.as<double>(), (int)sl.nsl, sl.step, seg.as<int64_t>(), nb_, spw,
                                                  npart.as<double>(), tab.as<double>(), s));
                }
                LPVS_TRY(launch_ap_assemble_fourier(tab.as<double>(), sd.eps.as<double>(), Nf, sl.s0, sl.delta, (int)zf, nreg, Gdst, np, nb_, sl.nsl * 4,
                                                    np * np, s));                                       // Q = A'WA   src/lasso.jl:119
                tr.mark("gram (structured)");
                for (int64_t q = 0; q < ns; ++q) {   // q_s = A'W y_s for every signal sharing the window   :120
                    if (wnufft_rhs) {
                        if (q >= 1 && q % 2 == 1)        // signals 1, 2 | 3, 4 | ... two to a pass
                            LPVS_TRY(launch_nufft_window_spread(dt.p, Wd, dys[(size_t)q].p, q + 1 < ns ? dys[(size_t)q + 1].p : nullptr, q + 1 < ns, offs.as<int64_t>(), nb_,
                                                                n, sl.step.hi[1], sl.step.lo[1], nfg, wgrids.as<double>() + (1 + q) * 2 * (int64_t)nfg,
                                                                wgrids.as<double>() + (2 + q) * 2 * (int64_t)nfg, gstride, s));
                        LPVS_TRY(launch_nufft_window_modes(wgrids.as<double>() + (1 + q) * 2 * (int64_t)nfg, gstride, nb_, nfg, (int)(sl.s0 / 2), (int)sl.nf8,
                                                           wscale_r.as<double>(), tabb.as<double>(), sl.nf8 * 4, s));
                    } else {
                        LPVS_TRY(launch_nudft_windows(dt.p, dys[(size_t)q].p, Wd, sd.rhi.as<double>(), sd.rlo.as<double>(), (int)sl.nf8, sl.step, seg.as<int64_t>(),
                                                      nb_, spw, npart.as<double>(), tabb.as<double>(), s));
                    }
                    LPVS_TRY(launch_ap_rhs_fourier(tabb.as<double>(), wnufft_rhs ? wepsr.as<double>() : sd.eps.as<double>(), Nf, (int)zf, bdst + q * np, nb_,
                                                   sl.nf8 * 4, ns * np, s));
                }
            } else {
                LPVS_TRY(launch_window_panels(dt.p, offs.as<int64_t>(), nb_, n, nrows, df.p, Nf, (int)zf, P.as<double>(), ld, s));
                tr.mark("panels");
                LPVS_TRY(launch_gram_panel_batch(pl, nb_, P.as<double>(), nrows * ld, ld, Wd, slab.as<double>(), s));
                tr.mark("gram");
                LPVS_HIP(hipMemsetAsync(Gdst, 0, sizeof(double) * (size_t)np * (size_t)np * (size_t)nb_, s));
                LPVS_HIP(hipMemsetAsync(bdst, 0, vb, s));
                LPVS_TRY(launch_gram_reduce_batch(pl, nb_, slab.as<double>(), Gdst, np, s));                  // Q = A'WA   src/lasso.jl:119
                for (int64_t q = 0; q < ns; ++q)
                    LPVS_TRY(launch_rhs_panel_batch(nb_, P.as<double>(), nrows * ld, ld, nreg, Wd, dys[(size_t)q].p, offs.as<int64_t>(), n,
                                                    bdst + q * np, ns * np, scr.as<double>(), scr.bytes, s));   // q = A'Wy   :120
            }
            return LPVS_OK;
        };
        LPVS_TRY(build_gram(Wdev, G, bvec.as<double>()));
        if (init) {
            // x0 = fourier_solve(A, y, zerofreq, lam) = (A'A + lam^2 I) \ A'y per window and signal -- the UNWEIGHTED problem, whatever W is
            // (src/lasso.jl:112 passes A and y, not Wd): with a window function its own Gram / right-hand sides, otherwise those just built
            if (Wdev != nullptr) LPVS_TRY(build_gram(nullptr, Q0.as<double>(), b0.as<double>()));
            else {
                LPVS_HIP(hipMemcpyAsync(Q0.p, G, sizeof(double) * (size_t)np * (size_t)np * (size_t)nb_, hipMemcpyDeviceToDevice, s));
                LPVS_HIP(hipMemcpyAsync(b0.p, bvec.p, sizeof(double) * (size_t)np * (size_t)nprob, hipMemcpyDeviceToDevice, s));
            }
            LPVS_HIP(hipMemcpyAsync(M0.p, Q0.p, sizeof(double) * (size_t)np * (size_t)np * (size_t)nb_, hipMemcpyDeviceToDevice, s));
            LPVS_TRY(launch_add_diag_batch(M0.as<double>(), np, nreg, a.lam * a.lam, nb_, s));
            LPVS_TRY(spd_inverse_inplace_batch(M0.as<double>(), np, nb_, work.as<double>(), istat.as<int>(), s));
            LPVS_HIP(hipMemcpyAsync(hist_.data(), istat.p, sizeof(int) * (size_t)nb_, hipMemcpyDeviceToHost, s));
            LPVS_HIP(hipStreamSynchronize(s));
            for (int q = 0; q < nb_; ++q)
                if (hist_[q] != 0) { set_error("window %lld: init = true needs (A'A + %.3g I) positive definite", (long long)(win_lo + w0 + q), a.lam * a.lam); return LPVS_ENUMERIC; }
            LPVS_TRY(launch_batch_ridge_solve(Q0.as<double>(), M0.as<double>(), np, nreg, nprob, (int)ns, b0.as<double>(), a.lam * a.lam, 2, x0buf.as<double>(),
                                              z.as<double>(), u.as<double>(), s));   // (z, u: scratch here; the ADMM init below rewrites them)
            tr.mark("init (ridge)");
        }
        LPVS_HIP(hipEventRecord(ev[0].b, s));
        if (sparse && a.linear_sign < 0) {  // Quadratic(Q, +q): the x-update's linear term is -q
            LPVS_HIP(hipMemcpyAsync(zh.data(), bvec.p, sizeof(double) * (size_t)np * (size_t)nprob, hipMemcpyDeviceToHost, s));
            LPVS_HIP(hipStreamSynchronize(s));
            for (size_t i = 0; i < (size_t)np * (size_t)nprob; ++i) zh[i] = -zh[i];
            LPVS_HIP(hipMemcpyAsync(bvec.p, zh.data(), sizeof(double) * (size_t)np * (size_t)nprob, hipMemcpyHostToDevice, s));
            LPVS_HIP(hipStreamSynchronize(s));
        }
        tr.mark("reduce+rhs");
        LPVS_HIP(hipEventRecord(ev[1].a, s));
        if (!sparse) LPVS_HIP(hipMemcpyAsync(M.p, Q.p, sizeof(double) * (size_t)np * (size_t)np * (size_t)nb_, hipMemcpyDeviceToDevice, s));
        const double shift = sparse ? 1.0 / mu : a.lam;       // (Q + I/mu) for the ADMM x-update; (A'WA + lam I) of src/lsfft.jl:77
        LPVS_TRY(launch_add_diag_batch(M.as<double>(), np, nreg, shift, nb_, s));
        LPVS_TRY(spd_inverse_inplace_batch(M.as<double>(), np, nb_, work.as<double>(), istat.as<int>(), s));
        LPVS_HIP(hipMemcpyAsync(hist_.data(), istat.p, sizeof(int) * (size_t)nb_, hipMemcpyDeviceToHost, s));
        LPVS_HIP(hipStreamSynchronize(s));
        for (int q = 0; q < nb_; ++q)
            if (hist_[q] != 0) { set_error("window %lld: (Q + %.3g I) is not positive definite", (long long)(win_lo + w0 + q), shift); return LPVS_ENUMERIC; }
        tr.mark("inverse");
        const double *sol = nullptr;
        if (sparse) {
            AdmmBatch ab{M.as<double>(), np, nreg, nprob, bvec.as<double>(), x.as<double>(), z.as<double>(), u.as<double>(), rhs.as<double>(),
                         mu, tol, a.prox_kind, a.prox_param, a.group_len, status.as<AdmmStatus>(), part.as<double>(), Mp.as<double>(), (int)ns};
            ab.opt_iteration = jopt[LPVS_OPT_ITERATION]; ab.opt_nt_loads = jopt[LPVS_OPT_NT_LOADS];
            ab.scratch = ballscr.p ? ballscr.as<double>() : nullptr;
            ab.x0 = init ? x0buf.as<double>() : nullptr;
            // 6-byte storage of the packed inverses + offset form of the x-update, as for the single problems (admm.hip); only
            // where the tile-packed path runs at all (LPVS_M_STORAGE=f64: doubles)
            const bool split = split_storage && admm_batch_uses_tiles(ab);
            if (split) {
                // mixed storage (36-bit fixed-point tiles where a window's inverse is small; the Fourier inverses are nearly diagonal):
                // tile formats and the per-matrix max|M| live behind the 6-byte slots of the Mp buffer (sized for doubles)
                const bool mixed = storage_opt != LPVS_STORAGE_SPLIT;
                const size_t nt = symv_packed_doubles(np) / (128 * 128);
                unsigned char *types = Mp.as<unsigned char>() + 6 * symv_packed_doubles(np) * (size_t)bw;
                const size_t types_bytes = ((nt * (size_t)bw + 255) / 256) * 256;
                if (mixed && ns == 1 && 6 * symv_packed_doubles(np) * (size_t)bw + types_bytes + 8 * (size_t)bw <= Mp.bytes) {
                    LPVS_TRY(launch_pack_tiles_mixed_batch(M.as<double>(), np, nb_, Mp.as<unsigned char>(), types,
                                                           reinterpret_cast<unsigned long long *>(types + types_bytes), s));
                    ab.mp_types = types;
                } else
                LPVS_TRY(launch_pack_tiles_split_batch(M.as<double>(), np, nb_, Mp.as<unsigned char>(), s));
                LPVS_TRY(launch_batch_matvec(M.as<double>(), np, nprob, (int)ns, bvec.as<double>(), xb.as<double>(), s));   // xb = M b, full precision
                ab.xb = xb.as<double>(); ab.mp_split = 1;
            } else {
                LPVS_TRY(launch_pack_tiles_batch(M.as<double>(), np, nb_, Mp.as<double>(), s));
            }
            LPVS_HIP(hipEventRecord(ev[1].b, s));
            LPVS_HIP(hipMemsetAsync(part.p, 0, part.bytes, s));   // tickets / block norms
            // one launch per iteration (admm.hip): accumulators / records of the whole pass, whether every tile is fixed point
            if (ab.mp_types != nullptr && ab.xb != nullptr && ns == 1) {
                if (!fibuf.p) LPVS_TRY(fibuf.alloc(sizeof(double) * fi_doubles(np, bw)));
                ab.fi = fibuf.as<double>();
                if (fi_batch_applicable(ab)) {
                    const size_t nt = symv_packed_doubles(np) / (128 * 128);
                    std::vector<unsigned char> ht(nt * (size_t)nb_);
                    LPVS_HIP(hipMemcpyAsync(ht.data(), ab.mp_types, ht.size(), hipMemcpyDeviceToHost, s));
                    LPVS_HIP(hipStreamSynchronize(s));
                    bool allfix = true;
                    for (unsigned char t : ht) allfix = allfix && t != 0;
                    ab.fi_prefetch_all = allfix ? 1 : 0;
                } else ab.fi = nullptr;
            }
            // 32-bit reads of the fixed-point tiles + the stale nibble product (admm.hip) where the batch iterates in one launch: the
            // default, and LPVS_STORAGE_MIXED32 by name; LPVS_STORAGE_MIXED reads all 36 bits.  (The refresh of a batch exists only inside the launch.)
            if (ab.fi != nullptr && (storage_opt == 0 || storage_opt == LPVS_STORAGE_MIXED32) && !(experiment_env("LPVS_NIB_FUSED") && experiment_env("LPVS_NIB_FUSED")[0] == '0')) {
                int period = 32, ramp = 8;
                if (const char *e = experiment_env("LPVS_NIB_PERIOD")) period = atoi(e) > 0 ? atoi(e) : 0;
                if (const char *e = experiment_env("LPVS_NIB_RAMP")) ramp = atoi(e) > 0 ? atoi(e) : 0;
                if (period > 0) {
                    if (!xbc.p) LPVS_TRY(xbc.alloc(vb));
                    if (!nibacc.p) LPVS_TRY(nibacc.alloc(vb));
                    LPVS_HIP(hipMemcpyAsync(xbc.p, xb.p, sizeof(double) * (size_t)np * (size_t)nprob, hipMemcpyDeviceToDevice, s));
                    LPVS_HIP(hipMemsetAsync(nibacc.p, 0, sizeof(double) * (size_t)np * (size_t)nprob, s));
                    ab.nib_period = period; ab.nib_ramp = ramp; ab.xb_corr = xbc.as<double>(); ab.nib_acc = nibacc.as<long long>();
                }
            }
            LPVS_HIP(hipEventRecord(ev[2].a, s));
            LPVS_TRY(launch_admm_batch_init(ab, s));
            if (ab.fi) LPVS_TRY(launch_fi_batch_setup(ab, s));
            for (int64_t done = 0; done < a.iters;) {   // chunks: stop early once every problem of the batch has converged
                const int64_t chunk = a.iters - done < 256 ? a.iters - done : 256;
                ab.fi_base = done;
                LPVS_TRY(launch_admm_batch_iterations(ab, chunk, s));
                done += chunk;
                LPVS_HIP(hipMemcpyAsync(hst.data(), status.p, sizeof(AdmmStatus) * (size_t)nprob, hipMemcpyDeviceToHost, s));
                LPVS_HIP(hipStreamSynchronize(s));
                bool all = true;
                for (int q = 0; q < nprob; ++q) all = all && hst[q].converged;
                if (all) break;
            }
            if (a.iters <= 0) {
                LPVS_HIP(hipMemcpyAsync(hst.data(), status.p, sizeof(AdmmStatus) * (size_t)nprob, hipMemcpyDeviceToHost, s));
                LPVS_HIP(hipStreamSynchronize(s));
            }
            LPVS_HIP(hipEventRecord(ev[2].b, s));
            g_win_timing[9] = ab.fi != nullptr ? (ab.nib_period > 0 ? 2 : 1) : 0;   // the pass ran one launch per iteration (2: reading 32 of the fixed-point tiles' 36 bits, stale nibble product)
            if (want_mv && w0 + nb_ >= nwin) {
                LPVS_TRY(launch_admm_batch_matvec_only(ab, 3, s));
                LPVS_HIP(hipEventRecord(ev[3].a, s));
                LPVS_TRY(launch_admm_batch_matvec_only(ab, 200, s));
                LPVS_HIP(hipEventRecord(ev[3].b, s));
                LPVS_HIP(hipStreamSynchronize(s));
                g_win_timing[4] = ev[3].ms() * 1e3 / 200; g_win_timing[5] = nb_;
                const size_t nt = symv_packed_doubles(np) / (128 * 128);
                double bytes = (double)(ab.mp_split ? 6 : 8) * (double)symv_packed_doubles(np) * (double)nb_;
                if (ab.mp_types) {
                    std::vector<unsigned char> ht(nt * (size_t)nb_);
                    LPVS_HIP(hipMemcpyAsync(ht.data(), ab.mp_types, ht.size(), hipMemcpyDeviceToHost, s));
                    LPVS_HIP(hipStreamSynchronize(s));
                    size_t nfix = 0, ndiag = 0;
                    for (unsigned char t : ht) { nfix += t != 0; ndiag += t == 2; }
                    bytes = (double)nfix * (double)(ab.nib_period > 0 ? kMixedFixed32TileBytes : kMixedFixedTileBytes) + (double)ndiag * 1024.0 + (double)(ht.size() - nfix) * (double)kMixedFloatTileBytes;   // (32-bit reads: no nibble planes)
                }
                g_win_timing[8] = bytes;
            }
            tr.mark("admm");
            sol = z.as<double>();
        } else {
            LPVS_HIP(hipEventRecord(ev[1].b, s));
            LPVS_HIP(hipEventRecord(ev[2].a, s));
            // x = (A'WA + lam I)^-1 A'W y, refined twice against the window's own Gram (src/lsfft.jl:77)
            LPVS_TRY(launch_batch_ridge_solve(Q.as<double>(), M.as<double>(), np, nreg, nprob, (int)ns, bvec.as<double>(), a.lam, 2, x.as<double>(),
                                              z.as<double>(), u.as<double>(), s));
            LPVS_HIP(hipEventRecord(ev[2].b, s));
            for (int q = 0; q < nprob; ++q) hst[q] = AdmmStatus{0, 1, 0, 0.0};
            tr.mark("ridge solves");
            sol = x.as<double>();
        }
        LPVS_HIP(hipMemcpyAsync(zh.data(), sol, sizeof(double) * (size_t)np * (size_t)nprob, hipMemcpyDeviceToHost, s));
        LPVS_HIP(hipStreamSynchronize(s));
        g_win_timing[0] += ev[0].ms(); g_win_timing[1] += ev[1].ms(); g_win_timing[2] += ev[2].ms(); g_win_timing[6] += 1;
        if (sparse && (a.st_x || a.st_z || a.st_u)) {   // the raw state of this pass's problems (lpvs_windows_estimate_state_f64)
            std::vector<double> sh((size_t)np * (size_t)nprob);
            const struct { const double *dev; double *out; } legs[3] = {{x.as<double>(), a.st_x}, {z.as<double>(), a.st_z}, {u.as<double>(), a.st_u}};
            for (const auto &lg : legs) {
                if (!lg.out) continue;
                LPVS_HIP(hipMemcpyAsync(sh.data(), lg.dev, sizeof(double) * sh.size(), hipMemcpyDeviceToHost, s));
                LPVS_HIP(hipStreamSynchronize(s));
                for (int q = 0; q < nb_; ++q)
                    for (int64_t sg = 0; sg < ns; ++sg)
                        std::memcpy(lg.out + ((size_t)sg * (size_t)a.st_nwin + (size_t)(win_lo + w0 + q - a.st_base)) * (size_t)nreg,
                                    sh.data() + ((size_t)q * (size_t)ns + (size_t)sg) * (size_t)np, sizeof(double) * (size_t)nreg);
            }
        }
        for (int q = 0; q < nb_; ++q)
            for (int64_t sg = 0; sg < ns; ++sg) {
                const double *c = zh.data() + ((size_t)q * (size_t)ns + (size_t)sg) * (size_t)np;   // fourier2complex, src/utilities.jl:62-73
                if (!zf) for (int64_t i = 0; i < Nf; ++i) { re[i] = c[i]; im[i] = c[Nf + i]; }
                else { re[0] = c[0]; im[0] = 0.0; for (int64_t i = 1; i < Nf; ++i) { re[i] = c[i]; im[i] = c[Nf + i - 1]; } }
                sink(w0 + q, sg, re.data(), im.data(), (int64_t)hst[(size_t)q * (size_t)ns + (size_t)sg].iters);
            }
    }
    return LPVS_OK;
}

// ---- the engine over a window range, in chunks that stay in the Infinity Cache, two parts of a chunk in flight ------------------------
// A sparse estimate streams every window's packed inverse once per ADMM iteration.  Measured at cfg4 (1024 windows, 0.74 MB of packed
// inverse each, 2000 iterations): all windows per launch 297 ms (HBM-bound: the 760 MB of a launch do not fit the 256 MiB Infinity
// Cache); chunks of 384 windows (285 MB), each chunk's two halves on two host threads / streams, 255 ms -- a chunk's inverses are
// re-read from the cache iteration after iteration, and one half's launch boundaries, ramp and tail (~4 us per launch) hide under
// the other's stream.  Chunks of 256 / 320 / 448 / 512 windows: 269 / 264 / 288 / 299 ms; three parts: 258 ms.  Small ranges (the
// shards of an 8-GPU run: 128 windows) gain the same way: 44.3 -> 36.7 ms.  The per-window results do not depend on how a range is
// cut (tests), and the sink sees them in window order -- buffered per part, replayed in order -- so every accumulation over windows
// is bit-identical to the uncut call.  LPVS_OPT_WINDOW_CHUNK_MB (default 1.0625 x the Infinity Cache = 285 MB; uncut: one chunk),
// LPVS_OPT_WINDOWS_IN_FLIGHT (default 2); the environment variables of the same names are the fallback.
template <class Sink>
int32_t windows_engine_chunked(const WinJob &a0, Sink sink) {
    const int64_t nwin = a0.win_hi - a0.win_lo;
    const bool sparse = a0.estimator == LPVS_EST_SPARSE || a0.estimator == LPVS_EST_SPARSE_INIT;
    // (options: the job's captured copy of its caller's defaults when it runs on a worker thread, else this thread's; then the environment)
    const int o_chunk = option_in_effect(LPVS_OPT_WINDOW_CHUNK_MB, a0.opt_captured ? a0.opt[LPVS_OPT_WINDOW_CHUNK_MB] : 0);
    const int o_fly = option_in_effect(LPVS_OPT_WINDOWS_IN_FLIGHT, a0.opt_captured ? a0.opt[LPVS_OPT_WINDOWS_IN_FLIGHT] : 0);
    const double chunk_mb = o_chunk == LPVS_WINDOW_UNCUT ? 0.0 : (o_chunk > 0 ? (double)o_chunk : 1.0625 * infinity_cache_bytes() * 1e-6);   // 285 MB for 256 MiB
    const int in_flight = o_fly > 0 ? o_fly : 2;
    int64_t zf = 0;
    if (!sparse || nwin < 16 || a0.iters < 64 || a0.freqs == nullptr || a0.Nf < 1 || lpvs_check_freq_f64(a0.freqs, a0.Nf, &zf) != LPVS_OK ||
        (in_flight == 1 && chunk_mb <= 0))
        return windows_engine(a0, sink);              // (argument errors are reported by the engine itself)
    const int64_t nreg = zf ? 2 * a0.Nf - 1 : 2 * a0.Nf, np = round_up(nreg, 128), nblk = np / 128;
    const double win_bytes = (double)(nblk * (nblk + 1) / 2) * (double)kMixedFixedTileBytes * (double)a0.ns;
    int64_t chunk = chunk_mb > 0 ? (int64_t)(chunk_mb * 1e6 / win_bytes) : nwin;
    if (chunk < 16) chunk = 16;
    if (chunk > nwin) chunk = nwin;
    const int64_t nchunks = ceil_div(nwin, chunk);
    chunk = ceil_div(nwin, nchunks);                  // even chunks
    WinJob a = a0;
    if (!a.opt_captured) { capture_default_options(a.opt); a.opt_captured = true; }   // (the parts run on other threads)
    a.f32_grid = a0.f32_grid || g_f32_admission;      // ... and so does the float-grid admission of the _f32 entry points (a thread-local switch)
    struct Rec { int64_t w, sg, its; std::vector<double> re, im; };
    double tsum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t Nf = a.Nf;
    for (int64_t c0 = 0; c0 < nwin; c0 += chunk) {
        const int64_t c1 = std::min(nwin, c0 + chunk), cw = c1 - c0;
        const int parts = cw >= 16 * in_flight ? in_flight : 1;
        std::vector<std::vector<Rec>> recs((size_t)parts);
        std::vector<int32_t> rcs((size_t)parts, LPVS_OK);
        std::vector<std::string> errs((size_t)parts);
        std::vector<std::array<double, 10>> tms((size_t)parts);
        auto run = [&](int p) {
            WinJob j = a;
            const int64_t lo = c0 + cw * p / parts, hi = c0 + cw * (p + 1) / parts;
            j.win_lo = a.win_lo + lo; j.win_hi = a.win_lo + hi;
            rcs[(size_t)p] = lpvs::windows_engine_run(j, [&](int64_t w, int64_t sg, const double *re, const double *im, int64_t its) {
                recs[(size_t)p].push_back(Rec{lo + w, sg, its, std::vector<double>(re, re + Nf), std::vector<double>(im, im + Nf)});
            });
            if (rcs[(size_t)p] != LPVS_OK) errs[(size_t)p] = lpvs_last_error();
            lpvs::windows_last_timing(tms[(size_t)p].data());
        };
        if (parts == 1) run(0);
        else {
            std::vector<std::thread> th;
            auto guarded = [&](int p) { run_guarded([&] { run(p); }, [&](int32_t rc) { rcs[(size_t)p] = rc; errs[(size_t)p] = lpvs_last_error(); }); };
            for (int p = 1; p < parts; ++p) th.emplace_back(guarded, p);
            guarded(0);
            for (auto &q : th) q.join();
        }
        for (int p = 0; p < parts; ++p)
            if (rcs[(size_t)p] != LPVS_OK) { set_error("%s", errs[(size_t)p].c_str()); return rcs[(size_t)p]; }
        for (int p = 0; p < parts; ++p)
            for (const Rec &r : recs[(size_t)p]) sink(r.w, r.sg, r.re.data(), r.im.data(), r.its);   // window order: parts are contiguous ranges
        double tmax[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = 0; p < parts; ++p)
            for (int i = 0; i < 10; ++i) {
                if (i == 0 || i == 1 || i == 2 || i == 4) tmax[i] = std::max(tmax[i], tms[(size_t)p][(size_t)i]);   // phases of the parts overlap
                else if (i == 7 || i == 9) tsum[i] = tms[(size_t)p][(size_t)i];
                else tsum[i] += tms[(size_t)p][(size_t)i];
            }
        for (int i : {0, 1, 2, 4}) tsum[i] += tmax[i];                                                                   // chunks follow each other
    }
    for (int i = 0; i < 10; ++i) g_win_timing[i] = tsum[i];
    return LPVS_OK;
}

// host-side output staging: results are assembled on the host and copied out once (outputs may be device pointers)
struct HostOut {
    double *user = nullptr; std::vector<double> stage; bool dev = false;
    void init(double *p, size_t count) { user = p; dev = p && is_device_ptr(p); if (dev) stage.assign(count, 0.0); }
    double *ptr() { return dev ? stage.data() : user; }
    int32_t finish() {
        if (dev) LPVS_HIP(hipMemcpy(user, stage.data(), sizeof(double) * stage.size(), hipMemcpyHostToDevice));
        return LPVS_OK;
    }
};

}  // namespace

namespace lpvs {
int32_t windows_engine_run(const WinJob &job, const WinSink &sink, bool chunked) {
    // (worker threads of multi.hip: the float-grid admission is a thread-local switch)
    struct Admit { bool prev; explicit Admit(bool on) : prev(g_f32_admission) { if (on) g_f32_admission = true; } ~Admit() { g_f32_admission = prev; } } admit(job.f32_grid);
    return chunked ? windows_engine_chunked(job, sink) : windows_engine(job, sink);
}
void windows_last_timing(double *out10) { for (int i = 0; i < 10; ++i) out10[i] = g_win_timing[i]; }
void windows_set_timing(const double *in10) { for (int i = 0; i < 10; ++i) g_win_timing[i] = in10[i]; }
void windows_set_multi_info(int rccl_ranks, int devices) { g_win_multi[0] = rccl_ranks; g_win_multi[1] = devices; }
}  // namespace lpvs

extern "C" {

int32_t lpvs_windowpsd_last_timing(double *out, int32_t n_out) {
    if (!out) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    for (int i = 0; i < n_out && i < 10; ++i) out[i] = g_win_timing[i];
    for (int i = 10; i < n_out && i < 12; ++i) out[i] = g_win_multi[i - 10];
    return LPVS_OK;
}

int32_t lpvs_windows_estimate_f64(const double *Y, int64_t ns, const double *t, int64_t L, int64_t n, int64_t noverlap, const double *W,
                                  const double *freqs, int64_t Nf, int32_t estimator, double lam, int32_t prox_kind, double prox_param,
                                  int64_t group_len, double mu, double tol, int64_t iters, int32_t linear_sign, int64_t win_lo,
                                  int64_t win_hi, int32_t device, double *x_re, double *x_im, int64_t *iters_out) {
    if (!Y || !t || !freqs || ns < 1) { set_error("NULL argument or ns < 1"); return LPVS_EARGUMENT; }
    std::vector<const double *> ys((size_t)ns);
    for (int64_t q = 0; q < ns; ++q) ys[(size_t)q] = Y + q * L;
    const WinJob job{ys.data(), ns, t, L, n, noverlap, W, freqs, Nf, estimator, lam, prox_kind, prox_param, group_len, mu, tol, iters, linear_sign,
                     win_lo, win_hi, device};
    const int64_t nwin = win_hi > win_lo ? win_hi - win_lo : 0;
    HostOut ore, oim;
    ore.init(x_re, (size_t)(ns * nwin * Nf)); oim.init(x_im, (size_t)(ns * nwin * Nf));
    double *pre = ore.ptr(), *pim = oim.ptr();
    LPVS_TRY(windows_engine_chunked(job, [&](int64_t w, int64_t sg, const double *re, const double *im, int64_t its) {
        if (pre) memcpy(pre + (size_t)(sg * nwin + w) * (size_t)Nf, re, sizeof(double) * (size_t)Nf);
        if (pim) memcpy(pim + (size_t)(sg * nwin + w) * (size_t)Nf, im, sizeof(double) * (size_t)Nf);
        if (iters_out) iters_out[sg * nwin + w] = its;
    }));
    LPVS_TRY(ore.finish());
    return oim.finish();
}

int32_t lpvs_windows_estimate_state_f64(const double *Y, int64_t ns, const double *t, int64_t L, int64_t n, int64_t noverlap, const double *W,
                                        const double *freqs, int64_t Nf, int32_t estimator, double lam, int32_t prox_kind, double prox_param,
                                        int64_t group_len, double mu, double tol, int64_t iters, int32_t linear_sign, int64_t win_lo,
                                        int64_t win_hi, int32_t device, double *x_out, double *z_out, double *u_out, int64_t *iters_out) {
    if (!Y || !t || !freqs || ns < 1) { set_error("NULL argument or ns < 1"); return LPVS_EARGUMENT; }
    if (estimator != LPVS_EST_SPARSE && estimator != LPVS_EST_SPARSE_INIT) { set_error("only the sparse estimators have an ADMM state"); return LPVS_EARGUMENT; }
    for (const double *p : {x_out, z_out, u_out})
        if (p && is_device_ptr(p)) { set_error("the state outputs are HOST arrays"); return LPVS_EARGUMENT; }
    std::vector<const double *> ys((size_t)ns);
    for (int64_t q = 0; q < ns; ++q) ys[(size_t)q] = Y + q * L;
    WinJob job{ys.data(), ns, t, L, n, noverlap, W, freqs, Nf, estimator, lam, prox_kind, prox_param, group_len, mu, tol, iters, linear_sign,
               win_lo, win_hi, device};
    const int64_t nwin = win_hi > win_lo ? win_hi - win_lo : 0;
    job.st_x = x_out; job.st_z = z_out; job.st_u = u_out; job.st_base = win_lo; job.st_nwin = nwin;
    LPVS_TRY(windows_engine_chunked(job, [&](int64_t w, int64_t sg, const double *, const double *, int64_t its) {
        if (iters_out) iters_out[(size_t)sg * (size_t)nwin + (size_t)w] = its;
    }));
    return LPVS_OK;
}

int32_t lpvs_windowpsd_sparse_f64(const double *y, const double *t, int64_t L, int64_t n, int64_t noverlap, const double *W,
                                  const double *freqs, int64_t Nf, int32_t prox_kind, double prox_param, int64_t group_len,
                                  double mu, double tol, int64_t iters, int32_t linear_sign, int64_t win_lo, int64_t win_hi,
                                  int32_t device, double *x_re, double *x_im, double *S_out, int64_t *iters_out) {
    const double *ys[1] = {y};
    const WinJob job{ys, 1, t, L, n, noverlap, W, freqs, Nf, LPVS_EST_SPARSE, 0.0, prox_kind, prox_param, group_len, mu, tol, iters, linear_sign,
                     win_lo, win_hi, device};
    const int64_t nwin = win_hi > win_lo ? win_hi - win_lo : 0;
    HostOut ore, oim, oS;
    ore.init(x_re, (size_t)(nwin * Nf)); oim.init(x_im, (size_t)(nwin * Nf)); oS.init(S_out, (size_t)Nf);
    double *pre = ore.ptr(), *pim = oim.ptr(), *pS = oS.ptr();
    if (pS) for (int64_t i = 0; i < Nf; ++i) pS[i] = 0.0;
    LPVS_TRY(windows_engine_chunked(job, [&](int64_t w, int64_t, const double *re, const double *im, int64_t its) {
        if (pS) for (int64_t i = 0; i < Nf; ++i) pS[i] += re[i] * re[i] + im[i] * im[i];   // S .+= abs2.(x), window order   src/lsfft.jl:122
        if (pre) memcpy(pre + (size_t)w * (size_t)Nf, re, sizeof(double) * (size_t)Nf);
        if (pim) memcpy(pim + (size_t)w * (size_t)Nf, im, sizeof(double) * (size_t)Nf);
        if (iters_out) iters_out[w] = its;
    }));
    LPVS_TRY(ore.finish()); LPVS_TRY(oim.finish());
    return oS.finish();
}

// ls_windowcsd / ls_cohere accumulators (src/lsfft.jl:140-156, :176-193): per window xy and xu from ONE Gram and ONE
// factorisation with two right-hand sides;  Syu += xy .* conj.(xu),  Syy += abs2.(xy),  Suu += abs2.(xu)  in window order
// (four real products per complex product, no fused multiply-add -- as Julia evaluates it).
int32_t lpvs_windowcsd_f64(const double *y, const double *u, const double *t, int64_t L, int64_t n, int64_t noverlap, const double *W,
                           const double *freqs, int64_t Nf, int32_t estimator, double lam, int32_t prox_kind, double prox_param,
                           int64_t group_len, double mu, double tol, int64_t iters, int32_t linear_sign, int64_t win_lo, int64_t win_hi,
                           int32_t device, double *Syu_re, double *Syu_im, double *Syy, double *Suu, double *x_re, double *x_im,
                           int64_t *iters_out) {
    if (!y || !u) { set_error("NULL signal"); return LPVS_EARGUMENT; }
    const double *ys[2] = {y, u};
    const WinJob job{ys, 2, t, L, n, noverlap, W, freqs, Nf, estimator, lam, prox_kind, prox_param, group_len, mu, tol, iters, linear_sign,
                     win_lo, win_hi, device};
    const int64_t nwin = win_hi > win_lo ? win_hi - win_lo : 0;
    HostOut ore, oim, o1, o2, o3, o4;
    ore.init(x_re, (size_t)(2 * nwin * Nf)); oim.init(x_im, (size_t)(2 * nwin * Nf));
    o1.init(Syu_re, (size_t)Nf); o2.init(Syu_im, (size_t)Nf); o3.init(Syy, (size_t)Nf); o4.init(Suu, (size_t)Nf);
    double *pre = ore.ptr(), *pim = oim.ptr(), *a1 = o1.ptr(), *a2 = o2.ptr(), *a3 = o3.ptr(), *a4 = o4.ptr();
    for (double *p : {a1, a2, a3, a4}) if (p) for (int64_t i = 0; i < Nf; ++i) p[i] = 0.0;
    std::vector<double> yr((size_t)Nf), yi((size_t)Nf);
    LPVS_TRY(windows_engine_chunked(job, [&](int64_t w, int64_t sg, const double *re, const double *im, int64_t its) {
        if (pre) memcpy(pre + (size_t)(sg * nwin + w) * (size_t)Nf, re, sizeof(double) * (size_t)Nf);
        if (pim) memcpy(pim + (size_t)(sg * nwin + w) * (size_t)Nf, im, sizeof(double) * (size_t)Nf);
        if (iters_out) iters_out[sg * nwin + w] = its;
        if (sg == 0) { memcpy(yr.data(), re, sizeof(double) * (size_t)Nf); memcpy(yi.data(), im, sizeof(double) * (size_t)Nf); return; }
        for (int64_t i = 0; i < Nf; ++i) {
            const double ar = yr[i], ai = yi[i], br = re[i], bi = im[i];
            if (a1) a1[i] += ar * br + ai * bi;            // Re xy conj(xu)
            if (a2) a2[i] += ai * br - ar * bi;            // Im xy conj(xu)
            if (a3) a3[i] += ar * ar + ai * ai;
            if (a4) a4[i] += br * br + bi * bi;
        }
    }));
    LPVS_TRY(ore.finish()); LPVS_TRY(oim.finish());
    LPVS_TRY(o1.finish()); LPVS_TRY(o2.finish()); LPVS_TRY(o3.finish());
    return o4.finish();
}

// ---- single-precision entry points (src/lasso.jl:85,91,144: the reference is eltype-generic) -------------------------
// Inputs are widened exactly to double, the assembly / Gram / factorisation run in double (at least as accurate as a
// Float32 run of the reference), the ADMM mat-vec of large problems streams a single-precision copy of M (half the
// bytes per iteration) with double accumulation, and outputs are rounded to float.
namespace {

// read-only float argument (host or device) widened into a device double buffer
struct WideArg {
    DevBuf buf;
    double *p = nullptr;
    int32_t set(const float *src, int64_t count, hipStream_t s) {
        if (src == nullptr || count <= 0) { p = nullptr; return LPVS_OK; }
        LPVS_TRY(buf.alloc(sizeof(double) * (size_t)count));
        p = buf.as<double>();
        if (is_device_ptr(src)) {
            LPVS_TRY(launch_cvt_f32_f64(src, p, count, s));
            LPVS_HIP(hipStreamSynchronize(s));
        } else {
            std::vector<double> h((size_t)count);
            for (int64_t i = 0; i < count; ++i) h[(size_t)i] = (double)src[i];
            LPVS_TRY(copy_to_device(p, h.data(), sizeof(double) * (size_t)count, s));
        }
        return LPVS_OK;
    }
};

// float output (host or device) produced from a device double buffer
int32_t narrow_out(float *dst, const double *src_dev, int64_t count, hipStream_t s) {
    if (dst == nullptr || count <= 0) return LPVS_OK;
    if (is_device_ptr(dst)) {
        LPVS_TRY(launch_cvt_f64_f32(src_dev, dst, count, s));
        LPVS_HIP(hipStreamSynchronize(s));
        return LPVS_OK;
    }
    std::vector<double> h((size_t)count);
    LPVS_TRY(copy_from_device(h.data(), src_dev, sizeof(double) * (size_t)count, s));
    for (int64_t i = 0; i < count; ++i) dst[i] = (float)h[(size_t)i];
    return LPVS_OK;
}

int32_t need_device() {
    if (lpvs_device_count() == 0) { set_error("no HIP device visible (the gfx950 path has no CPU fallback)"); return LPVS_EDEVICE; }
    return LPVS_OK;
}

}  // namespace

int32_t lpvs_check_freq_f32(const float *f, int64_t Nf, int64_t *zerofreq) {
    if (!f || Nf <= 0) { set_error("NULL argument or Nf <= 0"); return LPVS_EARGUMENT; }
    LPVS_TRY(need_device());
    WideArg df; LPVS_TRY(df.set(f, Nf, nullptr));
    return lpvs_check_freq_f64(df.p, Nf, zerofreq);
}

int32_t lpvs_fourier_regressor_f32(const float *t, int64_t N, const float *f, int64_t Nf, float *A_out, int64_t *zerofreq) {
    if (!t || !f || !A_out || N <= 0 || Nf <= 0) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    LPVS_TRY(need_device());
    WideArg dt, df; LPVS_TRY(dt.set(t, N, nullptr)); LPVS_TRY(df.set(f, Nf, nullptr));
    int64_t zf = 0;
    LPVS_TRY(lpvs_check_freq_f64(df.p, Nf, &zf));
    const int64_t nreg = zf ? 2 * Nf - 1 : 2 * Nf;
    DevBuf A; LPVS_TRY(A.alloc(sizeof(double) * (size_t)N * (size_t)nreg));
    LPVS_TRY(lpvs_fourier_regressor_f64(dt.p, N, df.p, Nf, A.as<double>(), zerofreq));
    return narrow_out(A_out, A.as<double>(), N * nreg, nullptr);
}

int32_t lpvs_lpv_regressor_f32(const float *X, const float *V, int64_t N, const float *w, int64_t Nf, int64_t Nv, int32_t normalize,
                               int32_t coulomb, int32_t permuted, float *Phi_out) {
    if (!X || !V || !w || !Phi_out || N <= 0 || Nf <= 0 || Nv <= 0) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    LPVS_TRY(need_device());
    WideArg dX, dV, dw; LPVS_TRY(dX.set(X, N, nullptr)); LPVS_TRY(dV.set(V, N, nullptr)); LPVS_TRY(dw.set(w, Nf, nullptr));
    const int64_t n = 2 * Nf * (coulomb ? 2 * Nv : Nv);
    DevBuf P; LPVS_TRY(P.alloc(sizeof(double) * (size_t)N * (size_t)n));
    LPVS_TRY(lpvs_lpv_regressor_f64(dX.p, dV.p, N, dw.p, Nf, Nv, normalize, coulomb, permuted, P.as<double>()));
    return narrow_out(Phi_out, P.as<double>(), N * n, nullptr);
}

int32_t lpvs_problem_create_fourier_f32(const float *y, const float *t, int64_t N, const float *f, int64_t Nf, const float *W,
                                        int32_t device, lpvs_problem **out) {
    if (!y || !t || !f || N <= 0 || Nf <= 0) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    LPVS_TRY(need_device());
    LPVS_HIP(hipSetDevice(device));
    WideArg dy, dt, df, dW;
    LPVS_TRY(dy.set(y, N, nullptr)); LPVS_TRY(dt.set(t, N, nullptr)); LPVS_TRY(df.set(f, Nf, nullptr)); LPVS_TRY(dW.set(W, N, nullptr));
    struct Admit { Admit() { g_f32_admission = true; } ~Admit() { g_f32_admission = false; } } admit;
    LPVS_TRY(lpvs_problem_create_fourier_f64(dy.p, dt.p, N, df.p, Nf, dW.p, device, out));
    (*out)->f32 = true;
    return LPVS_OK;
}

int32_t lpvs_problem_create_lpv_f32(const float *y, const float *X, const float *V, int64_t N, const float *w, int64_t Nf, int64_t Nv,
                                    int32_t normalize, int32_t coulomb, int32_t device, lpvs_problem **out) {
    if (!y || !X || !V || !w || N <= 0 || Nf <= 0) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    LPVS_TRY(need_device());
    LPVS_HIP(hipSetDevice(device));
    WideArg dy, dX, dV, dw;
    LPVS_TRY(dy.set(y, N, nullptr)); LPVS_TRY(dX.set(X, N, nullptr)); LPVS_TRY(dV.set(V, N, nullptr)); LPVS_TRY(dw.set(w, Nf, nullptr));
    struct Admit { Admit() { g_f32_admission = true; } ~Admit() { g_f32_admission = false; } } admit;
    LPVS_TRY(lpvs_problem_create_lpv_f64(dy.p, dX.p, dV.p, N, dw.p, Nf, Nv, normalize, coulomb, device, out));
    (*out)->f32 = true;
    return LPVS_OK;
}

int32_t lpvs_problem_create_lpv_multi_f32(const float *Y, int64_t ns, const float *X, const float *V, int64_t N, const float *w, int64_t Nf,
                                          int64_t Nv, int32_t normalize, int32_t coulomb, int32_t device, lpvs_problem **out) {
    if (!Y || !X || !V || !w || N <= 0 || Nf <= 0 || ns <= 0) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    LPVS_TRY(need_device());
    LPVS_HIP(hipSetDevice(device));
    WideArg dY, dX, dV, dw;
    LPVS_TRY(dY.set(Y, N * ns, nullptr)); LPVS_TRY(dX.set(X, N, nullptr)); LPVS_TRY(dV.set(V, N, nullptr)); LPVS_TRY(dw.set(w, Nf, nullptr));
    struct Admit { Admit() { g_f32_admission = true; } ~Admit() { g_f32_admission = false; } } admit;
    LPVS_TRY(lpvs_problem_create_lpv_multi_f64(dY.p, ns, dX.p, dV.p, N, dw.p, Nf, Nv, normalize, coulomb, device, out));
    (*out)->f32 = true;   // float I/O through the _f32 accessors; several right-hand sides keep the double matrix stream (matrix cores)
    return LPVS_OK;
}

// the batched-window engine for Float32 callers: float in / out, double arithmetic (float time stamps and frequencies are widened
// exactly; a float frequency grid is snapped to the progression it was rounded from, as in the other _f32 constructors)
int32_t lpvs_windows_estimate_f32(const float *Y, int64_t ns, const float *t, int64_t L, int64_t n, int64_t noverlap, const float *W,
                                  const float *freqs, int64_t Nf, int32_t estimator, double lam, int32_t prox_kind, double prox_param,
                                  int64_t group_len, double mu, double tol, int64_t iters, int32_t linear_sign, int64_t win_lo,
                                  int64_t win_hi, int32_t device, float *x_re, float *x_im, int64_t *iters_out) {
    if (!Y || !t || !freqs || ns < 1 || L <= 0 || Nf <= 0) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    LPVS_TRY(need_device());
    LPVS_HIP(hipSetDevice(device));
    WideArg dY, dt, dW, df;
    LPVS_TRY(dY.set(Y, L * ns, nullptr)); LPVS_TRY(dt.set(t, L, nullptr)); LPVS_TRY(dW.set(W, n, nullptr)); LPVS_TRY(df.set(freqs, Nf, nullptr));
    const int64_t nwin = win_hi > win_lo ? win_hi - win_lo : 0;
    const int64_t cnt = ns * nwin * Nf;
    DevBuf o; LPVS_TRY(o.alloc(sizeof(double) * (size_t)(cnt > 0 ? cnt : 1) * 2));
    struct Admit { Admit() { g_f32_admission = true; } ~Admit() { g_f32_admission = false; } } admit;
    LPVS_TRY(lpvs_windows_estimate_f64(dY.p, ns, dt.p, L, n, noverlap, dW.p, df.p, Nf, estimator, lam, prox_kind, prox_param, group_len, mu, tol,
                                       iters, linear_sign, win_lo, win_hi, device, o.as<double>(), o.as<double>() + cnt, iters_out));
    LPVS_TRY(narrow_out(x_re, o.as<double>(), cnt, nullptr));
    return narrow_out(x_im, o.as<double>() + cnt, cnt, nullptr);
}

// ls_windowcsd / ls_cohere accumulators for Float32 callers
int32_t lpvs_windowcsd_f32(const float *y, const float *u, const float *t, int64_t L, int64_t n, int64_t noverlap, const float *W,
                           const float *freqs, int64_t Nf, int32_t estimator, double lam, int32_t prox_kind, double prox_param,
                           int64_t group_len, double mu, double tol, int64_t iters, int32_t linear_sign, int64_t win_lo, int64_t win_hi,
                           int32_t device, float *Syu_re, float *Syu_im, float *Syy, float *Suu, float *x_re, float *x_im, int64_t *iters_out) {
    if (!y || !u || !t || !freqs || L <= 0 || Nf <= 0) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    LPVS_TRY(need_device());
    LPVS_HIP(hipSetDevice(device));
    WideArg dy, du, dt, dW, df;
    LPVS_TRY(dy.set(y, L, nullptr)); LPVS_TRY(du.set(u, L, nullptr)); LPVS_TRY(dt.set(t, L, nullptr)); LPVS_TRY(dW.set(W, n, nullptr));
    LPVS_TRY(df.set(freqs, Nf, nullptr));
    const int64_t nwin = win_hi > win_lo ? win_hi - win_lo : 0, cnt = 2 * nwin * Nf;
    DevBuf o; LPVS_TRY(o.alloc(sizeof(double) * (size_t)((cnt > 0 ? cnt : 1) * 2 + 4 * Nf)));
    double *xr = o.as<double>(), *xi = xr + cnt, *acc = xi + cnt;          // acc: Syu_re, Syu_im, Syy, Suu
    struct Admit { Admit() { g_f32_admission = true; } ~Admit() { g_f32_admission = false; } } admit;
    LPVS_TRY(lpvs_windowcsd_f64(dy.p, du.p, dt.p, L, n, noverlap, dW.p, df.p, Nf, estimator, lam, prox_kind, prox_param, group_len, mu, tol, iters,
                                linear_sign, win_lo, win_hi, device, acc, acc + Nf, acc + 2 * Nf, acc + 3 * Nf, x_re ? xr : nullptr,
                                x_im ? xi : nullptr, iters_out));
    LPVS_TRY(narrow_out(Syu_re, acc, Nf, nullptr)); LPVS_TRY(narrow_out(Syu_im, acc + Nf, Nf, nullptr));
    LPVS_TRY(narrow_out(Syy, acc + 2 * Nf, Nf, nullptr)); LPVS_TRY(narrow_out(Suu, acc + 3 * Nf, Nf, nullptr));
    LPVS_TRY(narrow_out(x_re, xr, cnt, nullptr));
    return narrow_out(x_im, xi, cnt, nullptr);
}

// one signal, sample rows sharded over ranks (see lpvs_problem_create_lpv_rows_f64), Float32 records
int32_t lpvs_problem_create_lpv_rows_f32(const float *Y, int64_t ns, const float *X, const float *V, int64_t N_local, const float *w, int64_t Nf,
                                         int64_t Nv, int32_t normalize, int32_t coulomb, const double *ranges4, int32_t device,
                                         lpvs_problem **out) {
    if (!Y || !X || !V || !w || N_local <= 0 || Nf <= 0 || ns <= 0) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    LPVS_TRY(need_device());
    LPVS_HIP(hipSetDevice(device));
    WideArg dY, dX, dV, dw;
    LPVS_TRY(dY.set(Y, N_local * ns, nullptr)); LPVS_TRY(dX.set(X, N_local, nullptr)); LPVS_TRY(dV.set(V, N_local, nullptr));
    LPVS_TRY(dw.set(w, Nf, nullptr));
    struct Admit { Admit() { g_f32_admission = true; } ~Admit() { g_f32_admission = false; } } admit;
    LPVS_TRY(lpvs_problem_create_lpv_rows_f64(dY.p, ns, dX.p, dV.p, N_local, dw.p, Nf, Nv, normalize, coulomb, ranges4, device, out));
    (*out)->f32 = true;
    return LPVS_OK;
}

// dense estimators on a Float32 handle: the ridge solution in floats
int32_t lpvs_problem_solve_ridge_f32(lpvs_problem *h, double ridge, float *x_out) {
    if (!h || !x_out) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    const int64_t cnt = h->n;                        // (single right-hand side, as the f64 entry point)
    DevBuf t; LPVS_TRY(t.alloc(sizeof(double) * (size_t)cnt));
    LPVS_TRY(lpvs_problem_solve_ridge_f64(h, ridge, t.as<double>()));
    return narrow_out(x_out, t.as<double>(), cnt, h->stream);
}

// re-entry from iterates saved as floats
int32_t lpvs_admm_set_state_f32(lpvs_problem *h, const float *x, const float *z, const float *u, int64_t iters_done) {
    if (!h || !x || !z || !u) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    const int64_t cnt = h->n * h->ns;
    WideArg dx, dz, du;
    LPVS_TRY(dx.set(x, cnt, h->stream)); LPVS_TRY(dz.set(z, cnt, h->stream)); LPVS_TRY(du.set(u, cnt, h->stream));
    return lpvs_admm_set_state_f64(h, dx.p, dz.p, du.p, iters_done);
}

int32_t lpvs_admm_init_f32(lpvs_problem *h, const float *x0, double mu, double tol, int32_t linear_sign) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    WideArg d0; LPVS_TRY(d0.set(x0, h->n * h->ns, h->stream));
    return lpvs_admm_init_f64(h, d0.p, mu, tol, linear_sign);
}

int32_t lpvs_admm_get_f32(lpvs_problem *h, float *x_out, float *z_out, float *u_out) {
    if (!h) { set_error("NULL handle"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    const int64_t cnt = h->n * h->ns;
    DevBuf t; LPVS_TRY(t.alloc(sizeof(double) * (size_t)cnt * 3));
    double *tx = t.as<double>(), *tz = tx + cnt, *tu = tz + cnt;
    LPVS_TRY(lpvs_admm_get_f64(h, x_out ? tx : nullptr, z_out ? tz : nullptr, u_out ? tu : nullptr));
    LPVS_TRY(narrow_out(x_out, tx, cnt, h->stream));
    LPVS_TRY(narrow_out(z_out, tz, cnt, h->stream));
    return narrow_out(u_out, tu, cnt, h->stream);
}

int32_t lpvs_problem_get_params_f32(lpvs_problem *h, int32_t which, float *re_out, float *im_out) {
    if (!h || !re_out || !im_out) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    LPVS_HIP(hipSetDevice(h->device));
    const int64_t m = (h->kind == 1 ? h->Nf * h->nb : h->Nf) * h->ns;
    DevBuf t; LPVS_TRY(t.alloc(sizeof(double) * (size_t)m * 2));
    LPVS_TRY(lpvs_problem_get_params_f64(h, which, t.as<double>(), t.as<double>() + m));
    LPVS_TRY(narrow_out(re_out, t.as<double>(), m, h->stream));
    return narrow_out(im_out, t.as<double>() + m, m, h->stream);
}

int32_t lpvs_ls_spectral_f32(const float *y, const float *t, int64_t N, const float *f, int64_t Nf, double lam, int32_t device,
                             float *re_out, float *im_out) {
    if (!y || !t || !f || !re_out || !im_out || N <= 0 || Nf <= 0) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    LPVS_TRY(need_device());
    LPVS_HIP(hipSetDevice(device));
    WideArg dy, dt, df; LPVS_TRY(dy.set(y, N, nullptr)); LPVS_TRY(dt.set(t, N, nullptr)); LPVS_TRY(df.set(f, Nf, nullptr));
    DevBuf o; LPVS_TRY(o.alloc(sizeof(double) * (size_t)Nf * 2));
    LPVS_TRY(lpvs_ls_spectral_f64(dy.p, dt.p, N, df.p, Nf, lam, device, o.as<double>(), o.as<double>() + Nf));
    LPVS_TRY(narrow_out(re_out, o.as<double>(), Nf, nullptr));
    return narrow_out(im_out, o.as<double>() + Nf, Nf, nullptr);
}

// ---- window bookkeeping (host integer arithmetic; src/windows.jl:27-36, :57-70) ----------------
int32_t lpvs_window_count(int64_t L, int64_t n, int64_t noverlap, int64_t *count) {
    if (!count) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    if (noverlap < 0) noverlap = n >> 1;  // src/windows.jl:29
    if (n <= 0 || noverlap >= n) { set_error("noverlap must be less than n"); return LPVS_EDOMAIN; }
    *count = L >= n ? (L - n) / (n - noverlap) + 1 : 0;
    return LPVS_OK;
}

int32_t lpvs_window_offsets(int64_t L, int64_t n, int64_t noverlap, int64_t *offsets, int64_t capacity, int64_t *count) {
    int64_t k = 0;
    LPVS_TRY(lpvs_window_count(L, n, noverlap, &k));
    if (noverlap < 0) noverlap = n >> 1;
    if (count) *count = k;
    if (offsets) {
        if (capacity < k) { set_error("offsets capacity %lld < %lld windows", (long long)capacity, (long long)k); return LPVS_EARGUMENT; }
        for (int64_t i = 0; i < k; ++i) offsets[i] = i * (n - noverlap);
    }
    return LPVS_OK;
}

int32_t lpvs_merge_f64(const double *yf, int64_t count, int64_t n, int64_t noverlap, int64_t L, double *ym) {
    if (!yf || !ym) { set_error("NULL argument"); return LPVS_EARGUMENT; }
    if (noverlap < 0) noverlap = n >> 1;
    if (n <= 0 || noverlap >= n) { set_error("noverlap must be less than n"); return LPVS_EDOMAIN; }
    std::vector<double> in, acc((size_t)L, 0.0);
    std::vector<int64_t> cnt((size_t)L, 0);
    LPVS_TRY(fetch_host(in, yf, count * n));
    int64_t lo = 0, hi = n - 1;
    for (int64_t w = 0; w < count; ++w) {
        for (int64_t i = lo; i <= hi && i < L; ++i) { acc[i] += in[w * n + (i - lo)]; cnt[i] += 1; }
        lo += n - noverlap; hi += n - noverlap;
        if (hi > L - 1) hi = L - 1;
    }
    for (int64_t i = 0; i < L; ++i) acc[i] /= (double)(cnt[i] > 1 ? cnt[i] : 1);
    if (is_device_ptr(ym)) { LPVS_HIP(hipMemcpy(ym, acc.data(), sizeof(double) * (size_t)L, hipMemcpyHostToDevice)); }
    else memcpy(ym, acc.data(), sizeof(double) * (size_t)L);
    return LPVS_OK;
}

}  // extern "C"
