// Multi-device acquisition behind the C ABI: one host process (a MATLAB session, say) drives N GPUs.
//
// the (PRN, Doppler-bin) cells are independent and a PRN's decision needs all of its
// bins, so the search shards by (signal, PRN) job.  Jobs are spread by cost (one B1C job at 99.375 MS/s
// is ~77 B2a jobs) with the LPT rule -- heaviest job to the currently lightest device.  Every device holds
// the whole IF block of each signal it has jobs of, runs its shard through bds_acq_run (results are zero outside
// the shard), and ONE all-reduce(SUM) of 3 x max_prn f64 per signal over the devices (RCCL over xGMI) leaves
// the complete acqResults on every device: x + 0 is exact, so the result is bit-identical to a single-device
// run.  Tracking needs no exchange (channels are independent; replicas only): bds_multi_ctx hands out the
// per-device contexts for it.
//
// RCCL is resolved with dlopen at the first multi-device call (no link-time dependency: a process that already
// carries PyTorch's copy keeps using that one; a single-device context never needs it).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "bds_internal.h"

namespace bds {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        // test-hooks build: BDS_RCCL_LIB names the library to load instead of the default search (the
        // test that a missing library is a clean BDS_ERR_UNSUPPORTED, tests/test_multi_gpu.py)
#ifdef BDS_TEST_HOOKS
        const char *forced = std::getenv("BDS_RCCL_LIB");
#else
        const char *forced = nullptr;  // the release library loads RCCL by its soname / from /opt/rocm/lib only
#endif
        std::string last = "?";
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (forced) name = forced;
            (void)dlerror();
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
            const char *e = dlerror();  // (read once: dlerror() clears the message, a second call returns NULL)
            if (e) last = e;
            if (forced) break;
        }
        if (!lib) {
            err = "cannot load RCCL: " + last;
            return false;
        }
#define BDS_SYM(f)                                                         \
    f = (decltype(f))dlsym(lib, "nccl" #f);                               \
    if (!f) {                                                              \
        err = "RCCL lacks nccl" #f;                                        \
        return false;                                                      \
    }
        BDS_SYM(CommInitAll)
        BDS_SYM(CommDestroy)
        BDS_SYM(AllReduce)
        BDS_SYM(GroupStart)
        BDS_SYM(GroupEnd)
        BDS_SYM(GetErrorString)
        BDS_SYM(CommCount)
#undef BDS_SYM
        return true;
    }
};

}  // namespace bds

struct bds_multi {
    std::vector<bds_ctx *> ctx;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm;   // one per device once RCCL is up
    std::vector<double *> d_buf;    // per device: 4 * BDS_MAX_PRN doubles (all-reduce buffer: 3 result rows + detected)
    bds::Rccl rccl;
    bool rccl_up = false;
    bool alias = false;             // BDS_MULTI_TEST_ALIAS: several contexts on one physical device, exchange summed on the host
    std::string err;
    int last_rccl_ranks = 0;        // communicator size reported by RCCL at the last all-reduce (diagnostics)
};

static thread_local std::string g_multi_create_error;

static int mfail(bds_multi *m, int code, const std::string &msg) {
    if (m)
        m->err = msg;
    else
        g_multi_create_error = msg;
    return code;
}

// ---- job partition -------------------------------------------------------------------------------------
// LPT (longest processing time first): jobs in decreasing cost order, each to the device with the least load so
// far (ties: lowest rank; equal costs keep their list order, so a single-signal list degenerates to round-robin).
extern "C" int bds_shard_jobs(int n_jobs, const double *cost, int world, int32_t *rank_of_job) {
    if (n_jobs < 0 || world < 1 || (n_jobs > 0 && (!cost || !rank_of_job))) return BDS_ERR_ARG;
    std::vector<int> order((size_t)n_jobs);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    std::vector<double> load((size_t)world, 0.0);
    for (int j : order) {
        int best = 0;
        for (int r = 1; r < world; ++r)
            if (load[(size_t)r] < load[(size_t)best]) best = r;
        rank_of_job[j] = best;
        load[(size_t)best] += cost[j];
    }
    return BDS_OK;
}

// Relative cost of searching ONE PRN with these settings: transform points x Doppler bins x components
// (the search dominates a call; forward transforms and refinement are a few percent).
extern "C" double bds_acq_job_cost(const bds_settings *s_in) {
    if (!s_in) return 0;
    bds_settings eff = *s_in;
    double new_fs, new_if, wp[2];
    if (bds_resample_plan(s_in, &new_fs, &new_if, wp) == 1) eff.samplingFreq = new_fs;
    const double spc = (double)bds::samples_per_code(eff);
    double N, X;
    int ncomp;
    if (eff.signal == BDS_SIGNAL_B1C) {
        X = bds::m_round(spc / 10 * eff.acqCohT);
        N = bds::m_round(spc / 10 * (10 + eff.acqCohT));
        ncomp = eff.pilotACQflag == 1 ? 2 : 1;
    } else {
        X = spc;
        N = 2 * spc;
        ncomp = 2;
    }
    const double D = bds::m_round(eff.acqSearchBand * 2 / eff.acqStep) + 1;
    return (N + X) * D * ncomp;
}

// ---- context ---------------------------------------------------------------------------------------------
extern "C" bds_multi *bds_multi_create(int n_devices, const int *device_ids) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        mfail(nullptr, BDS_ERR_HIP, "no HIP device visible: libbds_mi355x has no CPU fallback");
        return nullptr;
    }
    // Test hook BDS_MULTI_TEST_ALIAS=1 (a one-GPU box): device ids may repeat / wrap round the visible devices, so that the
    // thread-per-device partition, the concurrent contexts and the reassembly run with n > 1; RCCL refuses two ranks on
    // one device, so the exchange of such a handle is the same sum taken on the host.
    // (only the exact value "1" arms it, and it says so on stderr: a deployment that merely inherits the variable gets the
    //  duplicate-device check and RCCL as usual)
#ifdef BDS_TEST_HOOKS
    const char *alias_env = std::getenv("BDS_MULTI_TEST_ALIAS");
#else
    const char *alias_env = nullptr;  // (test-hooks build only)
#endif
    const bool alias = alias_env && std::strcmp(alias_env, "1") == 0;
    if (alias) fprintf(stderr, "[bds] BDS_MULTI_TEST_ALIAS=1: test hook active -- repeated device ids accepted, exchange summed on the host instead of RCCL\n");
    if (n_devices <= 0) n_devices = n;  // all visible devices
    if (n_devices > n && !device_ids && !alias) {
        mfail(nullptr, BDS_ERR_ARG, "bds_multi_create: " + std::to_string(n_devices) + " devices requested, " + std::to_string(n) + " visible");
        return nullptr;
    }
    bds_multi *m = new bds_multi();
    m->alias = alias;
    for (int i = 0; i < n_devices; ++i) {
        const int d = device_ids ? device_ids[i] : (alias ? i % n : i);
        if (!alias && std::find(m->dev.begin(), m->dev.end(), d) != m->dev.end()) {
            mfail(nullptr, BDS_ERR_ARG, "bds_multi_create: device " + std::to_string(d) + " listed twice");
            bds_multi_destroy(m);
            return nullptr;
        }
        bds_ctx *c = bds_create(d);
        if (!c) {
            mfail(nullptr, BDS_ERR_HIP, std::string("bds_multi_create: ") + bds_last_error(nullptr));
            bds_multi_destroy(m);
            return nullptr;
        }
        m->ctx.push_back(c);
        m->dev.push_back(d);
        double *p = nullptr;
        (void)hipSetDevice(d);
        if (hipMalloc((void **)&p, sizeof(double) * 4 * BDS_MAX_PRN) != hipSuccess) {
            mfail(nullptr, BDS_ERR_NOMEM, "bds_multi_create: hipMalloc of the all-reduce buffer failed");
            bds_multi_destroy(m);
            return nullptr;
        }
        m->d_buf.push_back(p);
    }
    return m;
}

extern "C" void bds_multi_destroy(bds_multi *m) {
    if (!m) return;
    for (size_t i = 0; i < m->comm.size(); ++i)
        if (m->comm[i]) (void)m->rccl.CommDestroy(m->comm[i]);
    for (size_t i = 0; i < m->d_buf.size(); ++i) {
        (void)hipSetDevice(m->dev[i]);
        (void)hipFree(m->d_buf[i]);
    }
    for (bds_ctx *c : m->ctx) bds_destroy(c);
    delete m;
}

extern "C" const char *bds_multi_last_error(const bds_multi *m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }
extern "C" int bds_multi_size(const bds_multi *m) { return m ? (int)m->ctx.size() : 0; }
extern "C" bds_ctx *bds_multi_ctx(bds_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[(size_t)i] : nullptr; }
extern "C" int bds_multi_rccl_ranks(const bds_multi *m) { return m ? m->last_rccl_ranks : 0; }

static int rccl_bring_up(bds_multi *m) {
    if (m->rccl_up) return BDS_OK;
    if (!m->rccl.load()) return mfail(m, BDS_ERR_UNSUPPORTED, m->rccl.err);
    m->comm.assign(m->ctx.size(), nullptr);
    const ncclResult_t r = m->rccl.CommInitAll(m->comm.data(), (int)m->dev.size(), m->dev.data());
    if (r != ncclSuccess) return mfail(m, BDS_ERR_HIP, std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(r));
    m->rccl_up = true;
    return BDS_OK;
}

// all-reduce(SUM) of `count` doubles held in every device's d_buf, in place, on the contexts' streams
static int allreduce_sum(bds_multi *m, size_t count) {
    int rc = rccl_bring_up(m);
    if (rc) return rc;
    const int n = (int)m->ctx.size();
    ncclResult_t r = m->rccl.GroupStart();
    for (int i = 0; i < n && r == ncclSuccess; ++i) {
        (void)hipSetDevice(m->dev[(size_t)i]);
        r = m->rccl.AllReduce(m->d_buf[(size_t)i], m->d_buf[(size_t)i], count, ncclDouble, ncclSum, m->comm[(size_t)i],
                              (hipStream_t)m->ctx[(size_t)i]->stream);
    }
    const ncclResult_t r2 = m->rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return mfail(m, BDS_ERR_HIP, std::string("ncclAllReduce: ") + m->rccl.GetErrorString(r));
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(m->dev[(size_t)i]);
        const hipError_t e = hipStreamSynchronize((hipStream_t)m->ctx[(size_t)i]->stream);
        if (e != hipSuccess) return mfail(m, BDS_ERR_HIP, std::string("all-reduce stream: ") + hipGetErrorString(e));
    }
    int ranks = 0;
    (void)m->rccl.CommCount(m->comm[0], &ranks);
    m->last_rccl_ranks = ranks;
    return BDS_OK;
}

// ---- acquisition over all devices -----------------------------------------------------------------------
extern "C" int bds_acquire_multi(bds_multi *m, int n_sig, const bds_acq_job *sig) {
    if (!m || n_sig < 1 || !sig) return BDS_ERR_ARG;
    const int world = (int)m->ctx.size();
    // (signal, PRN) job list with per-job cost
    struct Job {
        int sig, prn;
    };
    std::vector<Job> jobs;
    std::vector<double> cost;
    for (int i = 0; i < n_sig; ++i) {
        const bds_acq_job &g = sig[i];
        if (!g.settings || !g.samples || !g.carrFreq || !g.codePhase || !g.peakMetric || g.max_prn < 1 || g.max_prn > BDS_MAX_PRN)
            return mfail(m, BDS_ERR_ARG, "bds_acquire_multi: signal " + std::to_string(i) + " has a NULL field or max_prn out of 1..63");
        const double c = bds_acq_job_cost(g.settings);
        std::vector<int> seen;
        for (int k = 0; k < g.settings->n_acq; ++k) {
            const int p = g.settings->acqSatelliteList[k];
            if (p > g.max_prn) return mfail(m, BDS_ERR_ARG, "bds_acquire_multi: max_prn < max(acqSatelliteList)");
            if (std::find(seen.begin(), seen.end(), p) != seen.end()) continue;  // a repeated PRN is one job
            seen.push_back(p);
            jobs.push_back({i, p});
            cost.push_back(c);
        }
    }
    std::vector<int32_t> rank_of(jobs.size());
    int rc = bds_shard_jobs((int)jobs.size(), cost.data(), world, rank_of.data());
    if (rc) return mfail(m, rc, "bds_shard_jobs failed");
    // per device and signal: its PRN shard, then the partial results (zero outside the shard)
    std::vector<std::vector<double>> part((size_t)world * n_sig);  // [dev][sig] -> 3 * max_prn (+ detected as 4th row)
    std::vector<int> dev_rc((size_t)world, BDS_OK);
    std::vector<std::string> dev_err((size_t)world);
    auto work = [&](int d) {
      try {  // (an exception leaving a std::thread is std::terminate: a failed allocation must come back as a status)
        bds_ctx *c = m->ctx[(size_t)d];
        for (int i = 0; i < n_sig; ++i) {
            const bds_acq_job &g = sig[i];
            std::vector<int32_t> shard;
            for (size_t j = 0; j < jobs.size(); ++j)
                if (jobs[j].sig == i && rank_of[j] == d) shard.push_back(jobs[j].prn);
            std::vector<double> &out = part[(size_t)d * n_sig + i];
            out.assign((size_t)4 * g.max_prn, 0.0);
            if (shard.empty()) continue;  // this device holds no job of the signal: contributes zeros
            std::vector<int32_t> det((size_t)g.max_prn, 0);
            int r = bds_acq_load(c, g.settings, g.samples, g.n_samples, g.is_complex);
            if (!r) r = bds_acq_prepare(c, g.settings);
            if (!r)
                r = bds_acq_run(c, g.settings, shard.data(), (int)shard.size(), g.max_prn, &out[0], &out[(size_t)g.max_prn],
                                &out[(size_t)2 * g.max_prn], det.data());
            if (r) {
                dev_rc[(size_t)d] = r;
                dev_err[(size_t)d] = "device " + std::to_string(m->dev[(size_t)d]) + ": " + bds_last_error(c);
                return;
            }
            for (int k = 0; k < g.max_prn; ++k) out[(size_t)3 * g.max_prn + k] = det[(size_t)k];
        }
      } catch (const std::bad_alloc &) {
        dev_rc[(size_t)d] = BDS_ERR_NOMEM;
        dev_err[(size_t)d] = "device " + std::to_string(m->dev[(size_t)d]) + ": out of host memory";
      } catch (const std::exception &e) {
        dev_rc[(size_t)d] = BDS_ERR_HIP;
        dev_err[(size_t)d] = "device " + std::to_string(m->dev[(size_t)d]) + ": " + e.what();
      }
    };
    if (world == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int d = 0; d < world; ++d) th.emplace_back(work, d);
        for (auto &t : th) t.join();
    }
    for (int d = 0; d < world; ++d)
        if (dev_rc[(size_t)d]) return mfail(m, dev_rc[(size_t)d], dev_err[(size_t)d]);
    // one all-reduce(SUM) per signal over the devices
    for (int i = 0; i < n_sig; ++i) {
        const bds_acq_job &g = sig[i];
        const size_t cnt = (size_t)4 * g.max_prn;
        std::vector<double> res(cnt);
        if (world == 1 && !m->ctx[0]->tune.multi_force_rccl) {
            res = part[(size_t)i];  // a single device has nothing to exchange
        } else if (m->alias) {
            // test hook: the all-reduce(SUM) on the host (contexts share a physical device)
            std::fill(res.begin(), res.end(), 0.0);
            for (int d = 0; d < world; ++d)
                for (size_t k = 0; k < cnt; ++k) res[k] += part[(size_t)d * n_sig + i][k];
        } else {
            for (int d = 0; d < world; ++d) {
                (void)hipSetDevice(m->dev[(size_t)d]);
                const hipError_t e = hipMemcpyAsync(m->d_buf[(size_t)d], part[(size_t)d * n_sig + i].data(), sizeof(double) * cnt,
                                                    hipMemcpyHostToDevice, (hipStream_t)m->ctx[(size_t)d]->stream);
                if (e != hipSuccess) return mfail(m, BDS_ERR_HIP, std::string("all-reduce staging: ") + hipGetErrorString(e));
            }
            if ((rc = allreduce_sum(m, cnt))) return rc;
            (void)hipSetDevice(m->dev[0]);
            const hipError_t e = hipMemcpy(res.data(), m->d_buf[0], sizeof(double) * cnt, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return mfail(m, BDS_ERR_HIP, std::string("all-reduce result: ") + hipGetErrorString(e));
        }
        for (int k = 0; k < g.max_prn; ++k) {
            g.carrFreq[k] = res[(size_t)k];
            g.codePhase[k] = res[(size_t)g.max_prn + k];
            g.peakMetric[k] = res[(size_t)2 * g.max_prn + k];
            if (g.detected) g.detected[k] = (int32_t)res[(size_t)3 * g.max_prn + k];
        }
    }
    return BDS_OK;
}
