// multi.hip -- one host process driving several MI355X of a node ("Process model").
//
// The windows of ls_windowpsd / ls_windowcsd / ls_cohere (the loop of src/lsfft.jl:120-123, :149-153, :183-190) are
// independent: contiguous window ranges go to the devices, one host thread + stream + engine pass per device, no
// data-path collective.  The per-window coefficients of all shards are then gathered with ONE RCCL all-gather over xGMI
// (equal-size slots; a few MB, latency-bound) so that device devices[0] holds every window, and are read back once; the
// caller accumulates |x|^2 / xy conj(xu) in window order, as the reference does.
//
// RCCL is bound at run time (dlopen of librccl.so.1): the library has no link-time dependency on it, a single-device
// build of the host language needs none, and a process that already carries an RCCL (PyTorch-ROCm bundles one) keeps one
// copy.  ngpus = 1 does not touch RCCL.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "lpvs_internal.h"

namespace lpvs {
namespace {

// the handful of RCCL entry points used (signatures of rccl.h 2.x; ncclDouble = 8, ncclSuccess = 0)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return lib && CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd; }
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl q;
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *nm : names) if (!q.lib) q.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);   // one already in the process?
        for (const char *nm : names) if (!q.lib) q.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (!q.lib) q.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (q.lib) {
            q.CommInitAll = reinterpret_cast<decltype(q.CommInitAll)>(dlsym(q.lib, "ncclCommInitAll"));
            q.CommDestroy = reinterpret_cast<decltype(q.CommDestroy)>(dlsym(q.lib, "ncclCommDestroy"));
            q.AllGather = reinterpret_cast<decltype(q.AllGather)>(dlsym(q.lib, "ncclAllGather"));
            q.GroupStart = reinterpret_cast<decltype(q.GroupStart)>(dlsym(q.lib, "ncclGroupStart"));
            q.GroupEnd = reinterpret_cast<decltype(q.GroupEnd)>(dlsym(q.lib, "ncclGroupEnd"));
            q.GetErrorString = reinterpret_cast<decltype(q.GetErrorString)>(dlsym(q.lib, "ncclGetErrorString"));
        }
        return q;
    }();
    return r;
}

// communicators are expensive to create (hundreds of ms): kept per device list for the life of the process
std::mutex g_comm_mu;
std::map<std::vector<int>, std::vector<void *>> g_comms;

int32_t comms_for(const std::vector<int> &devs, std::vector<void *> **out) {
    std::lock_guard<std::mutex> g(g_comm_mu);
    auto it = g_comms.find(devs);
    if (it == g_comms.end()) {
        Rccl &R = rccl();
        if (!R.ok()) { const char *de = dlerror(); set_error("ngpus > 1 needs RCCL for the final gather, and librccl.so.1 could not be loaded: %s", de ? de : "symbols missing"); return LPVS_EDEVICE; }
        std::vector<void *> c(devs.size(), nullptr);
        const int rc = R.CommInitAll(c.data(), (int)devs.size(), devs.data());
        if (rc != 0) { set_error("ncclCommInitAll over %zu devices failed: %s", devs.size(), R.GetErrorString ? R.GetErrorString(rc) : "?"); return LPVS_EDEVICE; }
        it = g_comms.emplace(devs, std::move(c)).first;
    }
    *out = &it->second;
    return LPVS_OK;
}

struct Shard {
    int device = 0;
    int64_t lo = 0, hi = 0;
    std::vector<double> host;      // slot image: [window][signal][2 Nf + 1] (re, im, iterations)
    DevBuf send, recv;
    hipStream_t stream = nullptr;
    int32_t rc = LPVS_OK;
    std::string err;
    double timing[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

}  // namespace
}  // namespace lpvs

using namespace lpvs;

extern "C" {

static int32_t windows_estimate_multi(const double *Y, int64_t ns, const double *t, int64_t L, int64_t n, int64_t noverlap,
                                      const double *W, const double *freqs, int64_t Nf, int32_t estimator, double lam,
                                      int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol, int64_t iters,
                                      int32_t linear_sign, const int32_t *devices, int32_t ngpus, double *x_re, double *x_im,
                                      int64_t *iters_out, bool f32_grid) {
    if (!Y || !t || !freqs || ns < 1 || Nf < 1) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); set_error("no HIP device visible (the gfx950 path has no CPU fallback)"); return LPVS_EDEVICE; }
    // the calling thread's current device is the caller's business (PyTorch, AMDGPU.jl): whatever this call selects, it puts back
    struct DeviceRestore { int dev = -1; DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; } }
                           ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); } } restore_device;
    if (ngpus <= 0) ngpus = count;                               // all visible devices
    if (ngpus > count && devices == nullptr) { set_error("ngpus = %d but %d device(s) visible", ngpus, count); return LPVS_EDEVICE; }
    std::vector<int> devs((size_t)ngpus);
    bool shared = false;
    for (int r = 0; r < ngpus; ++r) {
        devs[(size_t)r] = devices ? devices[r] : r;
        if (devs[(size_t)r] < 0 || devs[(size_t)r] >= count) { set_error("device %d out of range [0,%d)", devs[(size_t)r], count); return LPVS_EDEVICE; }
        for (int q = 0; q < r; ++q)
            if (devs[(size_t)q] == devs[(size_t)r]) {
                // rehearsal on a box with fewer GPUs than shards (tests): shards may share a device, and then the gather cannot be
                // an RCCL collective (one rank per device) -- the shard images are taken from the host staging instead
                if (getenv("LPVS_MULTI_ALLOW_SHARED_DEVICE") == nullptr) { set_error("device %d listed twice", devs[(size_t)r]); return LPVS_EARGUMENT; }
                shared = true;
            }
    }
    // LPVS_MULTI_FORCE_RCCL: also a single device goes through the (one-rank) all-gather -- exercises the RCCL binding on a 1-GPU box
    const bool use_rccl = !shared && (ngpus > 1 || getenv("LPVS_MULTI_FORCE_RCCL") != nullptr);
    int64_t k = 0;
    LPVS_TRY(lpvs_window_count(L, n, noverlap, &k));
    if (k == 0) return LPVS_OK;
    // arguments that live on a device are staged through the host once: every device thread uploads what it needs
    std::vector<double> hY, ht, hW, hf;
    auto host_of = [&](const double *p, int64_t cnt, std::vector<double> &stage) -> const double * {
        if (!p || !is_device_ptr(p)) return p;
        stage.resize((size_t)cnt);
        if (hipMemcpy(stage.data(), p, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return stage.data();
    };
    const double *Yh = host_of(Y, L * ns, hY), *th = host_of(t, L, ht), *Wh = host_of(W, n, hW), *fh = host_of(freqs, Nf, hf);
    if (!Yh || !th || !fh || (W && !Wh)) { set_error("staging of device arguments failed"); return LPVS_EDEVICE; }
    double tam = 0;
    for (int64_t i = 0; i < L; ++i) { const double a = th[i] < 0 ? -th[i] : th[i]; if (a > tam) tam = a; }   // one admission decision for all shards

    const int64_t cap = (k + ngpus - 1) / ngpus;                 // equal-size slots keep the gather a single all-gather
    const size_t per_win = (size_t)ns * (size_t)(2 * Nf + 1), slot = (size_t)cap * per_win;
    std::vector<Shard> sh((size_t)ngpus);
    std::vector<const double *> ys((size_t)ns);
    for (int64_t q = 0; q < ns; ++q) ys[(size_t)q] = Yh + q * L;
    const int64_t base = k / ngpus, rem = k % ngpus;
    for (int r = 0; r < ngpus; ++r) {
        Shard &S = sh[(size_t)r];
        S.device = devs[(size_t)r];
        S.lo = r * base + (r < rem ? r : rem);
        S.hi = S.lo + base + (r < rem ? 1 : 0);
        S.host.assign(slot, 0.0);
    }
    int copt[kOptCount];
    capture_default_options(copt);                               // the caller's default options travel to the worker threads with the job
    auto work = [&](int r) {
        Shard &S = sh[(size_t)r];
        if (hipSetDevice(S.device) != hipSuccess) { (void)hipGetLastError(); S.rc = LPVS_EDEVICE; S.err = "hipSetDevice failed"; return; }
        WinJob job{ys.data(), ns, th, L, n, noverlap, Wh, fh, Nf, estimator, lam, prox_kind, prox_param, group_len, mu, tol, iters, linear_sign,
                   S.lo, S.hi, S.device};
        job.t_absmax = tam;
        job.f32_grid = f32_grid;
        for (int i = 0; i < kOptCount; ++i) job.opt[i] = copt[i];
        job.opt_captured = true;
        double *img = S.host.data();
        S.rc = windows_engine_run(job, [&](int64_t w, int64_t sg, const double *re, const double *im, int64_t its) {
            double *e = img + ((size_t)w * (size_t)ns + (size_t)sg) * (size_t)(2 * Nf + 1);
            memcpy(e, re, sizeof(double) * (size_t)Nf);
            memcpy(e + Nf, im, sizeof(double) * (size_t)Nf);
            e[2 * Nf] = (double)its;
        }, /*chunked=*/true);
        if (S.rc != LPVS_OK) { S.err = lpvs_last_error(); return; }
        windows_last_timing(S.timing);
        if (use_rccl) {   // the shard's slot goes to its device for the gather
            if ((S.rc = S.send.alloc(sizeof(double) * slot)) != LPVS_OK || (S.rc = S.recv.alloc(sizeof(double) * slot * (size_t)ngpus)) != LPVS_OK) { S.err = lpvs_last_error(); return; }
            if (hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking) != hipSuccess ||
                hipMemcpyAsync(S.send.p, img, sizeof(double) * slot, hipMemcpyHostToDevice, S.stream) != hipSuccess ||
                hipStreamSynchronize(S.stream) != hipSuccess) { (void)hipGetLastError(); S.rc = LPVS_EDEVICE; S.err = "upload of the shard's coefficients failed"; }
        }
    };
    if (ngpus == 1) work(0);
    else {
        std::vector<std::thread> th_;
        for (int r = 0; r < ngpus; ++r)
            th_.emplace_back([&, r] { run_guarded([&] { work(r); }, [&](int32_t rc) { sh[(size_t)r].rc = rc; sh[(size_t)r].err = lpvs_last_error(); }); });
        for (auto &q : th_) q.join();
    }
    struct Cleanup { std::vector<Shard> &s; ~Cleanup() { for (auto &S : s) if (S.stream) { (void)hipSetDevice(S.device); (void)hipStreamSynchronize(S.stream); (void)hipStreamDestroy(S.stream); } } } cleanup{sh};
    for (auto &S : sh) if (S.rc != LPVS_OK) { set_error("device %d (windows [%lld,%lld)): %s", S.device, (long long)S.lo, (long long)S.hi, S.err.c_str()); return S.rc; }
    {   // timing of the call = the slowest shard's phases
        double tmax[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (auto &S : sh) for (int i = 0; i < 10; ++i) if (S.timing[i] > tmax[i]) tmax[i] = S.timing[i];
        tmax[3] = (double)k;
        windows_set_timing(tmax);
        windows_set_multi_info(0, ngpus);
    }
    std::vector<double> all;
    const double *img_all = nullptr;
    if (use_rccl) {
        std::vector<void *> *comms = nullptr;
        LPVS_TRY(comms_for(devs, &comms));
        Rccl &R = rccl();
        // nothing returns between GroupStart and GroupEnd: an open group would swallow the process's next collective
        int rc = R.GroupStart();
        bool dev_ok = true;
        for (int r = 0; r < ngpus && rc == 0 && dev_ok; ++r) {
            if (hipSetDevice(sh[(size_t)r].device) != hipSuccess) { (void)hipGetLastError(); dev_ok = false; break; }
            rc = R.AllGather(sh[(size_t)r].send.p, sh[(size_t)r].recv.p, slot, 8 /* ncclDouble */, (*comms)[(size_t)r], sh[(size_t)r].stream);
        }
        const int rc2 = R.GroupEnd();
        if (!dev_ok) { set_error("hipSetDevice failed while enqueueing the RCCL all-gather"); return LPVS_EDEVICE; }
        if (rc != 0 || rc2 != 0) { set_error("RCCL all-gather of the window coefficients failed: %s", R.GetErrorString ? R.GetErrorString(rc ? rc : rc2) : "?"); return LPVS_EDEVICE; }
        for (auto &S : sh) { LPVS_HIP(hipSetDevice(S.device)); LPVS_HIP(hipStreamSynchronize(S.stream)); }
        windows_set_multi_info((int)comms->size(), ngpus);          // the communicator that gathered: one rank per device
        all.resize(slot * (size_t)ngpus);
        LPVS_HIP(hipSetDevice(sh[0].device));
        LPVS_HIP(hipMemcpy(all.data(), sh[0].recv.p, sizeof(double) * all.size(), hipMemcpyDeviceToHost));   // devices[0] holds every window
        img_all = all.data();
    }
    // unpack in window order: x_re / x_im are ns x k x Nf (signal-major), iters_out ns x k
    const bool dre = x_re && is_device_ptr(x_re), dim_ = x_im && is_device_ptr(x_im);
    std::vector<double> sre, sim;
    if (dre) sre.resize((size_t)(ns * k * Nf));
    if (dim_) sim.resize((size_t)(ns * k * Nf));
    double *pre = dre ? sre.data() : x_re, *pim = dim_ ? sim.data() : x_im;
    for (int r = 0; r < ngpus; ++r) {
        const Shard &S = sh[(size_t)r];
        const double *img = use_rccl ? img_all + (size_t)r * slot : S.host.data();
        for (int64_t w = S.lo; w < S.hi; ++w)
            for (int64_t sg = 0; sg < ns; ++sg) {
                const double *e = img + ((size_t)(w - S.lo) * (size_t)ns + (size_t)sg) * (size_t)(2 * Nf + 1);
                if (pre) memcpy(pre + (size_t)(sg * k + w) * (size_t)Nf, e, sizeof(double) * (size_t)Nf);
                if (pim) memcpy(pim + (size_t)(sg * k + w) * (size_t)Nf, e + Nf, sizeof(double) * (size_t)Nf);
                if (iters_out) iters_out[sg * k + w] = (int64_t)e[2 * Nf];
            }
    }
    if (dre) LPVS_HIP(hipMemcpy(x_re, sre.data(), sizeof(double) * sre.size(), hipMemcpyHostToDevice));
    if (dim_) LPVS_HIP(hipMemcpy(x_im, sim.data(), sizeof(double) * sim.size(), hipMemcpyHostToDevice));
    return LPVS_OK;
}

int32_t lpvs_windows_estimate_multi_f64(const double *Y, int64_t ns, const double *t, int64_t L, int64_t n, int64_t noverlap,
                                        const double *W, const double *freqs, int64_t Nf, int32_t estimator, double lam,
                                        int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol, int64_t iters,
                                        int32_t linear_sign, const int32_t *devices, int32_t ngpus, double *x_re, double *x_im,
                                        int64_t *iters_out) {
    return windows_estimate_multi(Y, ns, t, L, n, noverlap, W, freqs, Nf, estimator, lam, prox_kind, prox_param, group_len, mu, tol, iters,
                                  linear_sign, devices, ngpus, x_re, x_im, iters_out, false);
}

// ---- multichannel LPV batches over several devices (SURVEY.md section 8(b)(4) "lpvs_lpv_batch", BASELINE.json config 5) ------------------
// ns channels sharing (X, V, w): contiguous channel ranges go to the devices, one host thread per device builds ITS shard's handle
// (one Gram / factorisation per device, every ADMM kernel advances the shard's channels in one pass over the inverse), runs the
// iterations and writes the shard's columns of the outputs.  No data-path collective; the coefficients (ns x Nf*Nv complex) come back
// through each device's own copy engine -- a host-process API returns host arrays, there is nothing to gather onto one device first.
int32_t lpvs_lpv_batch_multi_f64(const double *Y, int64_t ns, const double *X, const double *V, int64_t N, const double *w, int64_t Nf,
                                 int64_t Nv, int32_t normalize, int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol,
                                 int64_t iters, const int32_t *devices, int32_t ngpus, double *re_out, double *im_out, int64_t *iters_out) {
    if (!Y || !X || !V || !w || !re_out || !im_out || ns < 1 || N < 1 || Nf < 1 || Nv < 1) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    if (is_device_ptr(re_out) || is_device_ptr(im_out)) { set_error("lpvs_lpv_batch_multi: outputs are host arrays"); return LPVS_EARGUMENT; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); set_error("no HIP device visible (the gfx950 path has no CPU fallback)"); return LPVS_EDEVICE; }
    struct DeviceRestore { int dev = -1; DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; } }
                           ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); } } restore_device;
    if (ngpus <= 0) ngpus = count;
    if (ngpus > count && devices == nullptr) { set_error("ngpus = %d but %d device(s) visible", ngpus, count); return LPVS_EDEVICE; }
    if ((int64_t)ngpus > ns) ngpus = (int32_t)ns;                 // no more shards than channels
    std::vector<int> devs((size_t)ngpus);
    for (int r = 0; r < ngpus; ++r) {
        devs[(size_t)r] = devices ? devices[r] : r;
        if (devs[(size_t)r] < 0 || devs[(size_t)r] >= count) { set_error("device %d out of range [0,%d)", devs[(size_t)r], count); return LPVS_EDEVICE; }
    }
    std::vector<double> hY, hX, hV, hw;
    auto host_of = [&](const double *p, int64_t cnt, std::vector<double> &stage) -> const double * {
        if (!is_device_ptr(p)) return p;
        stage.resize((size_t)cnt);
        if (hipMemcpy(stage.data(), p, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return stage.data();
    };
    const double *Yh = host_of(Y, N * ns, hY), *Xh = host_of(X, N, hX), *Vh = host_of(V, N, hV), *wh = host_of(w, Nf, hw);
    if (!Yh || !Xh || !Vh || !wh) { set_error("staging of device arguments failed"); return LPVS_EDEVICE; }
    const int64_t m = Nf * Nv;                                    // complex parameters per channel
    struct ChShard { int device; int64_t lo, hi; int32_t rc = LPVS_OK; std::string err; };
    std::vector<ChShard> sh((size_t)ngpus);
    const int64_t base = ns / ngpus, rem = ns % ngpus;
    for (int r = 0; r < ngpus; ++r) {
        sh[(size_t)r].device = devs[(size_t)r];
        sh[(size_t)r].lo = r * base + (r < rem ? r : rem);
        sh[(size_t)r].hi = sh[(size_t)r].lo + base + (r < rem ? 1 : 0);
    }
    int copt[kOptCount];
    capture_default_options(copt);
    auto work = [&](int r) {
        ChShard &S = sh[(size_t)r];
        const int64_t cnt = S.hi - S.lo;
        for (int o = 1; o < kOptCount; ++o) (void)lpvs_set_default_option(o, copt[o]);   // the caller's default options on this worker thread
        lpvs_problem *h = nullptr;
        auto fail = [&](int32_t rc) { S.rc = rc; S.err = lpvs_last_error(); if (h) lpvs_problem_destroy(h); };
        int32_t rc = lpvs_problem_create_lpv_multi_f64(Yh + S.lo * N, cnt, Xh, Vh, N, wh, Nf, Nv, normalize, 0, S.device, &h);
        if (rc != LPVS_OK) return fail(rc);
        if ((rc = lpvs_problem_set_prox(h, prox_kind, prox_param, group_len)) != LPVS_OK) return fail(rc);
        if ((rc = lpvs_admm_init_f64(h, nullptr, mu, tol, LPVS_LINEAR_LEAST_SQUARES)) != LPVS_OK) return fail(rc);
        int64_t done = 0; double nxz = 0; int32_t conv = 0;
        if ((rc = lpvs_admm_run(h, iters, &done, &nxz, &conv)) != LPVS_OK) return fail(rc);
        if ((rc = lpvs_problem_get_params_f64(h, 0, re_out + S.lo * m, im_out + S.lo * m)) != LPVS_OK) return fail(rc);
        if (iters_out)
            for (int64_t q = 0; q < cnt; ++q) {
                int64_t it = 0;
                if ((rc = lpvs_admm_status(h, q, &it, nullptr, nullptr)) != LPVS_OK) return fail(rc);
                iters_out[S.lo + q] = it;
            }
        lpvs_problem_destroy(h);
    };
    if (ngpus == 1) {
        int saved[kOptCount];
        capture_default_options(saved);
        work(0);
        for (int o = 1; o < kOptCount; ++o) (void)lpvs_set_default_option(o, saved[o]);
    } else {
        std::vector<std::thread> th_;
        for (int r = 0; r < ngpus; ++r)
            th_.emplace_back([&, r] { run_guarded([&] { work(r); }, [&](int32_t rc) { sh[(size_t)r].rc = rc; sh[(size_t)r].err = lpvs_last_error(); }); });
        for (auto &q : th_) q.join();
    }
    for (auto &S : sh) if (S.rc != LPVS_OK) { set_error("device %d (channels [%lld,%lld)): %s", S.device, (long long)S.lo, (long long)S.hi, S.err.c_str()); return S.rc; }
    return LPVS_OK;
}

// ---- independent LPV signals over several devices, several in flight per device (BASELINE.json config 3 as a batch) ---------------------
// nsig signals, each with its OWN samples of X and V (so nothing is shared: one Gram, one factorisation, one ADMM run per signal) --
// the loop  [ls_sparse_spectral_lpv(Y[:,q], X[:,q], V[:,q], w, Nv; ...) for q]  of a host program.  Contiguous signal ranges go to the
// devices; per device `in_flight` host threads pull signals from the device's range, each solve on its own handle and stream: the
// matrix-core-bound Gram / factorisation of one solve runs under the HBM-bound iterations of another (cfg3 on one MI355X: 13.9 signals/s
// one at a time, 16.4-16.8 with two in flight, nothing more with three).  No collective.
int32_t lpvs_lpv_signals_multi_f64(const double *Y, const double *X, const double *V, int64_t N, int64_t nsig, const double *w, int64_t Nf,
                                   int64_t Nv, int32_t normalize, int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol,
                                   int64_t iters, const int32_t *devices, int32_t ngpus, int32_t in_flight, double *re_out, double *im_out,
                                   int64_t *iters_out) {
    if (!Y || !X || !V || !w || !re_out || !im_out || nsig < 1 || N < 1 || Nf < 1 || Nv < 1) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    if (is_device_ptr(re_out) || is_device_ptr(im_out)) { set_error("lpvs_lpv_signals_multi: outputs are host arrays"); return LPVS_EARGUMENT; }
    if (in_flight < 1 || in_flight > 8) { set_error("in_flight = %d: 1 .. 8 solves per device", in_flight); return LPVS_EARGUMENT; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); set_error("no HIP device visible (the gfx950 path has no CPU fallback)"); return LPVS_EDEVICE; }
    struct DeviceRestore { int dev = -1; DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; } }
                           ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); } } restore_device;
    if (ngpus <= 0) ngpus = count;
    if (ngpus > count && devices == nullptr) { set_error("ngpus = %d but %d device(s) visible", ngpus, count); return LPVS_EDEVICE; }
    if ((int64_t)ngpus > nsig) ngpus = (int32_t)nsig;
    std::vector<int> devs((size_t)ngpus);
    for (int r = 0; r < ngpus; ++r) {
        devs[(size_t)r] = devices ? devices[r] : r;
        if (devs[(size_t)r] < 0 || devs[(size_t)r] >= count) { set_error("device %d out of range [0,%d)", devs[(size_t)r], count); return LPVS_EDEVICE; }
    }
    // (device inputs are used in place when they live on the solving device; the constructors stage anything else themselves)
    const int64_t m = Nf * Nv;                                    // complex parameters per signal
    struct Range { int device; int64_t lo, hi; std::atomic<int64_t> next{0}; };
    std::vector<Range> rg((size_t)ngpus);
    const int64_t base = nsig / ngpus, rem = nsig % ngpus;
    for (int r = 0; r < ngpus; ++r) {
        rg[(size_t)r].device = devs[(size_t)r];
        rg[(size_t)r].lo = r * base + (r < rem ? r : rem);
        rg[(size_t)r].hi = rg[(size_t)r].lo + base + (r < rem ? 1 : 0);
        rg[(size_t)r].next.store(rg[(size_t)r].lo);
    }
    int copt[kOptCount];
    capture_default_options(copt);
    std::mutex err_mu;
    int32_t first_rc = LPVS_OK; std::string first_err; int64_t first_sig = -1;
    auto work = [&](int r) {
        Range &R = rg[(size_t)r];
        for (int o = 1; o < kOptCount; ++o) (void)lpvs_set_default_option(o, copt[o]);   // the caller's default options on this worker thread
        for (;;) {
            const int64_t q = R.next.fetch_add(1);
            if (q >= R.hi) return;
            { std::lock_guard<std::mutex> g(err_mu); if (first_rc != LPVS_OK) return; }
            lpvs_problem *h = nullptr;
            auto fail = [&](int32_t rc) {
                std::lock_guard<std::mutex> g(err_mu);
                if (first_rc == LPVS_OK) { first_rc = rc; first_err = lpvs_last_error(); first_sig = q; }
                if (h) lpvs_problem_destroy(h);
            };
            int32_t rc = lpvs_problem_create_lpv_f64(Y + q * N, X + q * N, V + q * N, N, w, Nf, Nv, normalize, 0, R.device, &h);
            if (rc != LPVS_OK) return fail(rc);
            if ((rc = lpvs_problem_set_prox(h, prox_kind, prox_param, group_len)) != LPVS_OK) return fail(rc);
            if ((rc = lpvs_admm_init_f64(h, nullptr, mu, tol, LPVS_LINEAR_LEAST_SQUARES)) != LPVS_OK) return fail(rc);
            int64_t done = 0; double nxz = 0; int32_t conv = 0;
            if ((rc = lpvs_admm_run(h, iters, &done, &nxz, &conv)) != LPVS_OK) return fail(rc);
            if ((rc = lpvs_problem_get_params_f64(h, 0, re_out + q * m, im_out + q * m)) != LPVS_OK) return fail(rc);
            if (iters_out) iters_out[q] = done;
            lpvs_problem_destroy(h);
        }
    };
    int saved[kOptCount];
    capture_default_options(saved);
    if (ngpus == 1 && in_flight == 1) work(0);
    else {
        std::vector<std::thread> th_;
        for (int r = 0; r < ngpus; ++r)
            for (int k = 0; k < in_flight; ++k)
                th_.emplace_back([&, r] { run_guarded([&] { work(r); }, [&](int32_t rc) {
                    std::lock_guard<std::mutex> g(err_mu);
                    if (first_rc == LPVS_OK) { first_rc = rc; first_err = lpvs_last_error(); first_sig = rg[(size_t)r].lo; } }); });
        for (auto &q : th_) q.join();
    }
    for (int o = 1; o < kOptCount; ++o) (void)lpvs_set_default_option(o, saved[o]);
    if (first_rc != LPVS_OK) { set_error("signal %lld: %s", (long long)first_sig, first_err.c_str()); return first_rc; }
    return LPVS_OK;
}

// ---- ls_windowpsd_lpv (src/lsfft.jl:267-277) inside the library ------------------------------------------------------------------
// Windows3(Y, X, V, n, noverlap, rect): every window is a dense LPV estimate of its own (ls_spectral_lpv, :239-250: basis centres from
// the WINDOW's range of V, its own Gram).  Two phases per chunk of windows (a chunk = what 24 GiB of matrices hold):
//   1. the windows' Grams G_q = Phi_q' Phi_q and right-hand sides Phi_q' [y_q, 1] by the single-problem builder (structured / NUFFT or
//      dense MFMA form, whatever its admission rules choose), `in_flight` windows at a time on host threads and streams of their
//      own, each written into its slot of ONE batch [windows][np][np];
//   2. the whole chunk through the batch machinery of the window engine: (G_q + lam^2 I)^-1 of all windows by ONE blocked sweep
//      (spd_inverse_inplace_batch), x_q = M_q b_q refined twice against every window's own Gram (launch_batch_ridge_solve), the
//      residual test of lpvs_problem_solve_ridge per window.
// S = sum over windows, IN WINDOW ORDER (:274), of abs2.(sum(reshape_params(x, Nf), dims = 2)).  fva_out (optional, k entries): the
// window's fraction of variance explained 1 - var(e)/var(y) (:255; the reference warns below 0.9 -- the bindings do) from the second
// right-hand side Phi'1.  A window whose normal equations are singular to working precision returns LPVS_ENUMERIC (the wrapper then
// takes the reference's QR route per window on the host).
int32_t lpvs_windowpsd_lpv_f64(const double *Y, const double *X, const double *V, int64_t N, const double *w, int64_t Nf, int64_t Nv,
                               int64_t n, int64_t noverlap, double lam, int32_t normalize, int32_t coulomb, int32_t device, int32_t in_flight,
                               double *S_out, double *fva_out) {
    if (!Y || !X || !V || !w || !S_out || N < 1 || Nf < 1 || Nv < 1 || n < 1) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    if (in_flight < 1 || in_flight > 8) { set_error("in_flight = %d: 1 .. 8 solves at a time", in_flight); return LPVS_EARGUMENT; }
    if (is_device_ptr(S_out) || (fva_out && is_device_ptr(fva_out))) { set_error("lpvs_windowpsd_lpv: S_out / fva_out are host arrays"); return LPVS_EARGUMENT; }
    int64_t k = 0;
    LPVS_TRY(lpvs_window_count(N, n, noverlap, &k));
    for (int64_t f = 0; f < Nf; ++f) S_out[f] = 0.0;
    if (k == 0) return LPVS_OK;
    std::vector<int64_t> offs((size_t)k);
    int64_t kk = 0;
    LPVS_TRY(lpvs_window_offsets(N, n, noverlap, offs.data(), k, &kk));
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); set_error("no HIP device visible (the gfx950 path has no CPU fallback)"); return LPVS_EDEVICE; }
    if (device < 0 || device >= count) { set_error("device %d out of range [0,%d)", device, count); return LPVS_EDEVICE; }
    struct DeviceRestore { int dev = -1; DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; } }
                           ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); } } restore_device;
    LPVS_HIP(hipSetDevice(device));
    const int64_t nb = coulomb ? 2 * Nv : Nv, m = Nf * nb, n2 = 2 * m, np = round_up(n2, 128);
    const int nrhs = 2;                                          // y and the ones vector (sum of the residuals for var(e))
    // the record on the host: y'y and sum(y) of every window, and the staging of [y_q, 1] for the constructor
    std::vector<double> Yh_;
    const double *Yh = Y;
    if (is_device_ptr(Y)) { Yh_.resize((size_t)N); LPVS_HIP(hipMemcpy(Yh_.data(), Y, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost)); Yh = Yh_.data(); }
    const bool xv_host = !is_device_ptr(X) && !is_device_ptr(V);
    const size_t mat = sizeof(double) * (size_t)np * (size_t)np;
    // windows per chunk: Qb + Mb + work + the eight [np][2] vectors of a window within HALF of what the device has free now (the rest
    // is for the in_flight handles building the Grams, each with its own G / state, and for whoever shares the device), 24 GiB at most;
    // halved again below if the allocation still fails.  A window's result depends on the chunking only to rounding: the batch
    // inverse takes the 64-wide sweep for two windows or more and the two-level schedule for a lone one.
    const size_t per_window = 2 * mat + spd_inverse_work_bytes(np) + 8 * sizeof(double) * (size_t)np * 2 + sizeof(int);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)48 << 30; }
    free_b += pool_cached_bytes(device);                           // blocks this library holds in its cache are free for this call
    size_t budget = free_b / 2 < ((size_t)24 << 30) ? free_b / 2 : ((size_t)24 << 30);
    int64_t cw = (int64_t)(budget / per_window);
    if (cw < 1) cw = 1;
    if (cw > k) cw = k;
    hipStream_t s = nullptr;
    LPVS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } sguard{s};
    DevBuf Qb, Mb, work, istat, bb, xv, t1, t2;
    DrainOnExit drain(s);
    for (;;) {                                                     // (out of memory with this chunk size: release, halve, try again)
        const size_t vb_ = sizeof(double) * (size_t)np * (size_t)cw * (size_t)nrhs;
        const int32_t rc = [&]() -> int32_t {
            LPVS_TRY(Qb.alloc(mat * (size_t)cw)); LPVS_TRY(Mb.alloc(mat * (size_t)cw));
            LPVS_TRY(work.alloc(spd_inverse_work_bytes(np) * (size_t)cw)); LPVS_TRY(istat.alloc(sizeof(int) * (size_t)cw));
            LPVS_TRY(bb.alloc(vb_)); LPVS_TRY(xv.alloc(vb_)); LPVS_TRY(t1.alloc(vb_)); LPVS_TRY(t2.alloc(vb_));
            return LPVS_OK;
        }();
        if (rc == LPVS_OK) break;
        if (rc != LPVS_ENOMEM || cw == 1) return rc;
        Qb.release(); Mb.release(); work.release(); istat.release(); bb.release(); xv.release(); t1.release(); t2.release();
        cw = (cw + 1) / 2;
    }
    std::vector<double> Sw((size_t)k * (size_t)Nf, 0.0);         // per-window contributions, summed in window order afterwards
    std::vector<double> hx((size_t)np * (size_t)cw * (size_t)nrhs), hq(hx.size()), hb(hx.size());
    std::vector<int> hist_((size_t)cw);
    int copt[kOptCount];
    capture_default_options(copt);
    int saved[kOptCount];
    capture_default_options(saved);
    struct RestoreOptions { int *saved; ~RestoreOptions() { for (int o = 1; o < kOptCount; ++o) (void)lpvs_set_default_option(o, saved[o]); } } ropt{saved};
    for (int64_t c0 = 0; c0 < k; c0 += cw) {
        const int nbw = (int)std::min<int64_t>(cw, k - c0);
        // ---- phase 1: the windows' Grams and right-hand sides into their slots
        std::atomic<int64_t> next{0};
        std::mutex err_mu;
        int32_t first_rc = LPVS_OK; std::string first_err; int64_t first_win = -1;
        auto fail_with = [&](int32_t rc, int64_t q) {
            std::lock_guard<std::mutex> g(err_mu);
            if (first_rc == LPVS_OK) { first_rc = rc; first_err = lpvs_last_error(); first_win = c0 + q; }
        };
        auto work_fn = [&]() {
            for (int o = 1; o < kOptCount; ++o) (void)lpvs_set_default_option(o, copt[o]);
            if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); set_error("hipSetDevice failed"); return fail_with(LPVS_EDEVICE, 0); }
            std::vector<double> y2((size_t)2 * (size_t)n, 1.0);   // [y_q | 1], column-major n x 2
            // (a stream of this thread's own for the copies into the batch: a plain hipMemcpy between device buffers may return before it
            // has run, and the handle's blocks go back to the pool right after it)
            hipStream_t cs = nullptr;
            if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); set_error("stream creation failed"); return fail_with(LPVS_EDEVICE, 0); }
            struct CsGuard { hipStream_t s; ~CsGuard() { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } csg{cs};
            for (;;) {
                const int64_t q = next.fetch_add(1);
                if (q >= nbw) return;
                { std::lock_guard<std::mutex> g(err_mu); if (first_rc != LPVS_OK) return; }
                const int64_t o = offs[(size_t)(c0 + q)];
                memcpy(y2.data(), Yh + o, sizeof(double) * (size_t)n);
                lpvs_problem *h = nullptr;
                int32_t rc;
                if (xv_host) {   // the window's ranges on the host: two device reductions and their synchronisations less per window
                    double r4[4] = {V[o], V[o], 0.0, 0.0};
                    bool bad = false;                              // a NaN fails every comparison below, wherever it sits: track it explicitly
                    for (int64_t i = 0; i < n; ++i) {
                        const double v = V[o + i], x = X[o + i], av = v < 0 ? -v : v, ax = x < 0 ? -x : x;
                        bad |= !(av < 0x1p1000) || !(ax < 0x1p1000);
                        if (v < r4[0]) r4[0] = v;
                        if (v > r4[1]) r4[1] = v;
                        if (av > r4[2]) r4[2] = av;
                        if (ax > r4[3]) r4[3] = ax;
                    }
                    if (bad || !(r4[0] <= r4[1])) {               // a NaN / Inf anywhere in the window: say so, instead of a "not positive definite" after a Gram build and a batch inverse
                        set_error("X or V holds a NaN or an Inf (samples %lld .. %lld)", (long long)o, (long long)(o + n - 1));
                        return fail_with(LPVS_EARGUMENT, q);
                    }
                    rc = lpvs_problem_create_lpv_rows_f64(y2.data(), nrhs, X + o, V + o, n, w, Nf, Nv, normalize, coulomb, r4, device, &h);
                } else rc = lpvs_problem_create_lpv_multi_f64(y2.data(), nrhs, X + o, V + o, n, w, Nf, Nv, normalize, coulomb, device, &h);
                if (rc != LPVS_OK) return fail_with(rc, q);
                double *G = nullptr, *b = nullptr; int64_t hnp = 0;
                rc = lpvs_problem_device_gram_f64(h, &G, &b, &hnp);
                if (rc == LPVS_OK && hnp != np) { set_error("window Gram of padded size %lld, expected %lld", (long long)hnp, (long long)np); rc = LPVS_EDEVICE; }
                if (rc == LPVS_OK && (hipMemcpyAsync(Qb.as<double>() + (size_t)q * (size_t)np * (size_t)np, G, mat, hipMemcpyDeviceToDevice, cs) != hipSuccess ||
                                      hipMemcpyAsync(bb.as<double>() + (size_t)q * (size_t)nrhs * (size_t)np, b, sizeof(double) * (size_t)np * (size_t)nrhs, hipMemcpyDeviceToDevice, cs) != hipSuccess ||
                                      hipStreamSynchronize(cs) != hipSuccess)) {
                    (void)hipGetLastError(); set_error("copy of a window's Gram into the batch failed"); rc = LPVS_EDEVICE;
                }
                lpvs_problem_destroy(h);
                if (rc != LPVS_OK) return fail_with(rc, q);
            }
        };
        const int nth = (int)std::min<int64_t>(in_flight, nbw);
        if (nth == 1) run_guarded(work_fn, [&](int32_t rc) { fail_with(rc, 0); });
        else {
            std::vector<std::thread> th_;
            for (int t = 0; t < nth; ++t) th_.emplace_back([&] { run_guarded(work_fn, [&](int32_t rc) { fail_with(rc, 0); }); });
            for (auto &q : th_) q.join();
        }
        if (first_rc != LPVS_OK) { set_error("window %lld: %s", (long long)first_win, first_err.c_str()); return first_rc; }
        LPVS_HIP(hipSetDevice(device));
        // ---- phase 2: all windows of the chunk at once
        const double ridge = lam * lam;                          // real_complex_bs(A, Y, lam): [A; lam I] \ [Y; 0], src/utilities.jl:49-54
        LPVS_HIP(hipMemcpyAsync(Mb.p, Qb.p, mat * (size_t)nbw, hipMemcpyDeviceToDevice, s));
        LPVS_TRY(launch_add_diag_batch(Mb.as<double>(), np, n2, ridge, nbw, s));
        LPVS_TRY(spd_inverse_inplace_batch(Mb.as<double>(), np, nbw, work.as<double>(), istat.as<int>(), s));
        LPVS_HIP(hipMemcpyAsync(hist_.data(), istat.p, sizeof(int) * (size_t)nbw, hipMemcpyDeviceToHost, s));
        LPVS_HIP(hipStreamSynchronize(s));
        for (int q = 0; q < nbw; ++q)
            if (hist_[(size_t)q] != 0) { set_error("window %lld: (G + %.3g I) is not positive definite to working precision", (long long)(c0 + q), ridge); return LPVS_ENUMERIC; }
        const int nprob = nbw * nrhs;
        LPVS_TRY(launch_batch_ridge_solve(Qb.as<double>(), Mb.as<double>(), np, n2, nprob, nrhs, bb.as<double>(), ridge, 2, xv.as<double>(), t1.as<double>(), t2.as<double>(), s));
        LPVS_TRY(launch_batch_matvec(Qb.as<double>(), np, nprob, nrhs, xv.as<double>(), t1.as<double>(), s));     // G x of the refined solutions
        const size_t cnt = (size_t)np * (size_t)nprob;
        LPVS_HIP(hipMemcpyAsync(hx.data(), xv.p, sizeof(double) * cnt, hipMemcpyDeviceToHost, s));
        LPVS_HIP(hipMemcpyAsync(hq.data(), t1.p, sizeof(double) * cnt, hipMemcpyDeviceToHost, s));
        LPVS_HIP(hipMemcpyAsync(hb.data(), bb.p, sizeof(double) * cnt, hipMemcpyDeviceToHost, s));
        LPVS_HIP(hipStreamSynchronize(s));
        for (int q = 0; q < nbw; ++q) {
            const double *x = hx.data() + (size_t)q * (size_t)nrhs * (size_t)np, *Gx = hq.data() + (size_t)q * (size_t)nrhs * (size_t)np;
            const double *b0 = hb.data() + (size_t)q * (size_t)nrhs * (size_t)np, *b1 = b0 + np;
            double r2 = 0, b2 = 0, xGx = 0, xb0 = 0, xb1 = 0;
            for (int64_t i = 0; i < n2; ++i) {
                const double r = b0[i] - (ridge * x[i] + Gx[i]);
                r2 += r * r; b2 += b0[i] * b0[i]; xGx += x[i] * Gx[i]; xb0 += x[i] * b0[i]; xb1 += x[i] * b1[i];
            }
            if (!(r2 <= 1e-18 * b2) && b2 > 0) {     // (as lpvs_problem_solve_ridge: the reference's QR of [A; lam I] still works there -- the wrapper's route)
                set_error("window %lld: normal equations (G + %.3g I) x = b are too ill-conditioned for the device solve (relative residual %.3g after refinement)",
                          (long long)(c0 + q), ridge, std::sqrt(r2 / b2));
                return LPVS_ENUMERIC;
            }
            double *sw = Sw.data() + (size_t)(c0 + q) * (size_t)Nf;
            for (int64_t f = 0; f < Nf; ++f) {                    // abs2(sum over the basis functions of frequency f): param f + (v-1) Nf = x[f 2nb + v] + i x[f 2nb + nb + v]
                double sr = 0.0, si = 0.0;
                for (int64_t v = 0; v < nb; ++v) { sr += x[f * 2 * nb + v]; si += x[f * 2 * nb + nb + v]; }
                sw[f] = sr * sr + si * si;
            }
            if (fva_out) {   // e = A x - y: |e|^2 = x'Gx - 2 x'b + y'y, sum(e) = x'(A'1) - sum(y); var with the n - 1 divisor (Statistics.var), src/lsfft.jl:252-255
                const double *yq = Yh + offs[(size_t)(c0 + q)];
                double yy = 0, ys = 0;
                for (int64_t i = 0; i < n; ++i) { yy += yq[i] * yq[i]; ys += yq[i]; }
                const double e2 = xGx - 2.0 * xb0 + yy, es = xb1 - ys;
                // (formed from separately rounded sums: good to ~1e-8 of var(y), which is all the 0.9 warning threshold of src/lsfft.jl:255 needs --
                // the header says so; a tiny negative var(e) from the cancellation is clamped, and a constant window (var(y) = 0), for which
                // the reference's 1 - var(e)/var(y) is -Inf or NaN, reports -Inf so that the bindings warn)
                double var_e = n > 1 ? (e2 - es * es / (double)n) / (double)(n - 1) : 0.0;
                const double var_y = n > 1 ? (yy - ys * ys / (double)n) / (double)(n - 1) : 0.0;
                if (var_e < 0) var_e = 0;
                fva_out[c0 + q] = var_y > 0 ? 1.0 - var_e / var_y : -HUGE_VAL;
            }
        }
    }
    for (int64_t q = 0; q < k; ++q)
        for (int64_t f = 0; f < Nf; ++f) S_out[f] += Sw[(size_t)q * (size_t)Nf + (size_t)f];                     // S += ..., window order (:274)
    return LPVS_OK;
}

// Float32 records (host or device): widened on the host, float grids snapped to their progression, coefficients returned as floats
int32_t lpvs_windows_estimate_multi_f32(const float *Y, int64_t ns, const float *t, int64_t L, int64_t n, int64_t noverlap,
                                        const float *W, const float *freqs, int64_t Nf, int32_t estimator, double lam,
                                        int32_t prox_kind, double prox_param, int64_t group_len, double mu, double tol, int64_t iters,
                                        int32_t linear_sign, const int32_t *devices, int32_t ngpus, float *x_re, float *x_im,
                                        int64_t *iters_out) {
    if (!Y || !t || !freqs || ns < 1 || Nf < 1 || L < 1) { set_error("NULL argument or empty size"); return LPVS_EARGUMENT; }
    auto widen = [&](const float *p, int64_t cnt, std::vector<double> &out) -> bool {
        if (!p) return true;
        std::vector<float> tmp;
        const float *src = p;
        if (is_device_ptr(p)) {
            tmp.resize((size_t)cnt);
            if (hipMemcpy(tmp.data(), p, sizeof(float) * (size_t)cnt, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return false; }
            src = tmp.data();
        }
        out.resize((size_t)cnt);
        for (int64_t i = 0; i < cnt; ++i) out[(size_t)i] = (double)src[i];
        return true;
    };
    std::vector<double> hY, ht, hW, hf;
    if (!widen(Y, L * ns, hY) || !widen(t, L, ht) || !widen(W, n, hW) || !widen(freqs, Nf, hf)) { set_error("staging of device arguments failed"); return LPVS_EDEVICE; }
    int64_t k = 0;
    LPVS_TRY(lpvs_window_count(L, n, noverlap, &k));
    const size_t cnt = (size_t)(ns * k * Nf);
    std::vector<double> re(cnt > 0 ? cnt : 1), im(cnt > 0 ? cnt : 1);
    LPVS_TRY(windows_estimate_multi(hY.data(), ns, ht.data(), L, n, noverlap, W ? hW.data() : nullptr, hf.data(), Nf, estimator, lam, prox_kind,
                                    prox_param, group_len, mu, tol, iters, linear_sign, devices, ngpus, re.data(), im.data(), iters_out, true));
    auto narrow = [&](float *dst, const std::vector<double> &v) -> bool {
        if (!dst || cnt == 0) return true;
        std::vector<float> f(cnt);
        for (size_t i = 0; i < cnt; ++i) f[i] = (float)v[i];
        return hipMemcpy(dst, f.data(), sizeof(float) * cnt, hipMemcpyDefault) == hipSuccess;
    };
    if (!narrow(x_re, re) || !narrow(x_im, im)) { (void)hipGetLastError(); set_error("copy of the coefficients failed"); return LPVS_EDEVICE; }
    return LPVS_OK;
}

}  // extern "C"
