// Library plumbing (version, per-thread error string) and the reward reduce kernel.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "tg_common.h"

static thread_local char g_err[512] = "";

void tg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* tg_last_error(void) { return g_err; }

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
namespace {
struct ProfRec { int id; hipEvent_t a, b; bool closed; unsigned long long serial; };
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
unsigned long long g_prof_serial = 0;
thread_local unsigned long long t_prof_open = 0;  // serial of the bracket this thread opened last (0 = none)
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

// A bracket is owned by the thread that opened it: tg_prof_end closes that thread's own record, so two threads
// profiling at once cannot pair each other's events.  A bracket an error path never closes is dropped (events
// recycled) by the next tg_prof_begin of the thread or by tg_profile_read.
void tg_prof_begin(int id, hipStream_t s) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{id, prof_event(), prof_event(), false, ++g_prof_serial};
    if (hipEventRecord(r.a, s) != hipSuccess) {
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
        t_prof_open = 0;
        return;
    }
    t_prof_open = r.serial;
    g_prof_recs.push_back(r);
}
void tg_prof_end(hipStream_t s) {
    if (!t_prof_open) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto it = g_prof_recs.rbegin(); it != g_prof_recs.rend(); ++it)
        if (it->serial == t_prof_open) {
            it->closed = hipEventRecord(it->b, s) == hipSuccess;
            break;
        }
    t_prof_open = 0;
}

extern "C" void tg_profile_enable(int on) { g_prof_on.store(on != 0); }

extern "C" int tg_profile_read(int id, double* total_ms, int* count) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double tot = 0.0;
    int n = 0, bad = 0;
    std::vector<ProfRec> keep;
    for (const ProfRec& r : g_prof_recs) {
        if (r.id != id) { keep.push_back(r); continue; }
        float ms = 0.f;
        // a record whose end event was never recorded (an entry point returned between begin and end) or whose
        // events cannot be queried is dropped, not kept to fail every later read
        if (r.closed && hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            tot += ms;
            ++n;
        } else {
            ++bad;
        }
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_recs.swap(keep);
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    if (bad) (void)hipGetLastError();  // clear the sticky error of a failed event query
    return TG_OK;
}
extern "C" const char* tg_version(void) { return "textgan_hip 0.4 gfx950"; }

// torch.mean(rewards.view(B, L, R), dim=-1) on the flat [R*L, B] buffer (utils/rollout.py:153):
// out[bt] = mean_r raw[bt*R + r].  One thread per output, sequential fp32 sum (R <= a few hundred).
__global__ __launch_bounds__(256) void k_reward_reduce(const float* __restrict__ raw, int BL, int R, float* __restrict__ out) {
    const int bt = blockIdx.x * 256 + threadIdx.x;
    if (bt >= BL) return;
    const float* p = raw + (size_t)bt * R;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += p[r];
    out[bt] = s / (float)R;
}

extern "C" int tg_reward_reduce(const float* raw, int B, int L, int R, float* out, void* stream) {
    if (B == 0 && L >= 1 && R >= 1) return TG_OK;   // an empty batch: no reward to write
    if (!raw || !out || B < 1 || L < 1 || R < 1) {
        tg_set_error("tg_reward_reduce: bad arguments (B=%d L=%d R=%d)", B, L, R);
        return TG_ERR_ARG;
    }
    const int BL = B * L;
    tg_prof_begin(TG_KERNEL_REDUCE, (hipStream_t)stream);
    hipLaunchKernelGGL(k_reward_reduce, dim3((BL + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, BL, R, out);
    tg_prof_end((hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tg_set_error("k_reward_reduce launch: %s", hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}

// ---- the one exchange of the path, for hosts that own an RCCL communicator themselves (C++ trainers; the Python
// shim uses torch.distributed's all_gather_into_tensor on its own process group instead, dist.py).  RCCL is not linked:
// ncclAllGather / ncclCommUserRank are taken from the copy of the library the process has already loaded (the one
// that created `comm`), so a communicator is never handed to a second copy.
namespace {
// Symbol from the RCCL copy already mapped into the process: the global scope first; else (torch loads librccl as a
// private dependency of libtorch_hip) the mapped file named librccl.so* in /proc/self/maps, re-opened with RTLD_NOLOAD
// -- which never loads anything, it only hands out a handle to what is there.
void* rccl_symbol(const char* name) {
    if (void* p = dlsym(RTLD_DEFAULT, name)) return p;
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return nullptr;
    char line[1024];
    void* sym = nullptr;
    while (!sym && fgets(line, sizeof(line), f)) {
        char* path = strchr(line, '/');
        if (!path) continue;
        path[strcspn(path, "\n")] = 0;
        const char* base = strrchr(path, '/') + 1;
        if (strncmp(base, "librccl.so", 10) != 0) continue;
        if (void* h = dlopen(path, RTLD_NOLOAD | RTLD_NOW)) {
            sym = dlsym(h, name);
            dlclose(h);  // drops the reference RTLD_NOLOAD took; the library stays mapped by its first loader
        }
    }
    fclose(f);
    return sym;
}
}  // namespace

extern "C" int tg_allgather_rows(float* raw, size_t rows_per_rank, size_t B, void* rccl_comm, void* stream) {
    if (!raw || !rccl_comm || rows_per_rank == 0 || B == 0) {
        tg_set_error("tg_allgather_rows: bad arguments (raw=%p comm=%p rows_per_rank=%zu B=%zu)", (void*)raw, rccl_comm,
                     rows_per_rank, B);
        return TG_ERR_ARG;
    }
    typedef int (*rank_fn)(void*, int*);
    typedef int (*gather_fn)(const void*, void*, size_t, int, void*, void*);
    // looked up per call: the library that owns `comm` may have been loaded after an earlier, failed call
    const rank_fn f_rank = reinterpret_cast<rank_fn>(rccl_symbol("ncclCommUserRank"));
    const gather_fn f_gather = reinterpret_cast<gather_fn>(rccl_symbol("ncclAllGather"));
    if (!f_rank || !f_gather) {
        tg_set_error("tg_allgather_rows: no RCCL in this process (ncclAllGather not found)");
        return TG_ERR_LAUNCH;
    }
    int rank = -1;
    if (f_rank(rccl_comm, &rank) != 0 || rank < 0) {
        tg_set_error("tg_allgather_rows: ncclCommUserRank failed");
        return TG_ERR_LAUNCH;
    }
    const size_t count = rows_per_rank * B;
    // ncclDataType_t of rccl.h (ROCm 7.2 rccl/rccl.h:466: ncclFloat32 = ncclFloat = 7; the value is part of the NCCL
    // ABI since 2.0 and RCCL keeps it) -- the header is not included so that no RCCL version is baked into the library
    const int kNcclFloat32 = 7;
    const int rc = f_gather(raw + (size_t)rank * count, raw, count, kNcclFloat32, rccl_comm, stream);  // in place
    if (rc != 0) {
        tg_set_error("tg_allgather_rows: ncclAllGather returned %d", rc);
        return TG_ERR_LAUNCH;
    }
    return TG_OK;
}
