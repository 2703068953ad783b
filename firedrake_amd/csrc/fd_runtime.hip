// fd_runtime.hip -- runtime half of the C ABI (include/fdhip.h): device memory, streams,
// events, and loading/launching the wrapper kernels that replace the JIT-compiled
// `wrap_<kernel>` of pyop2/global_kernel.py:426-456.
#include "fd_common.h"
#include <cstring>
#include <cstdlib>
#include <dlfcn.h>
#include <unistd.h>
#include <fcntl.h>
#include <spawn.h>
#include <sstream>
#include <sys/stat.h>
#include <sys/wait.h>
#include <map>
#include <mutex>
#include <vector>

extern char **environ;

namespace fd {
static hipStream_t g_default_stream = nullptr;
static hipStream_t g_saved_stream = nullptr;      // the default stream a graph capture displaced (fd_graph_begin .. fd_graph_end)
// hipFree synchronises the device and is illegal while this thread captures a stream: the capture is invalidated and every later
// launch fails with "a previous error during capture".  A release can arrive at any moment (a finaliser of the host language's
// garbage collector running inside the captured step), so releases between fd_graph_begin and fd_graph_end are parked here and
// carried out by fd_graph_end.
static bool g_capturing = false;
static std::vector<void *> g_deferred_free;
static std::vector<fd_kernel_t> g_deferred_kernels;            // (hipModuleUnload and the destruction of another graph's stream: the same)
static std::vector<fd_graph_t> g_deferred_graphs;
static std::vector<hipStream_t> g_deferred_streams;
hipError_t release(void *p) {
    if (!p) return hipSuccess;
    if (g_capturing) { g_deferred_free.push_back(p); return hipSuccess; }
    return hipFree(p);
}
hipStream_t default_stream() { return g_default_stream; }
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

struct Builtin { const char *name; const void *fn; };
static std::vector<Builtin> &registry() { static std::vector<Builtin> r; return r; }
int register_builtin(const char *name, const void *fn) { registry().push_back({name, fn}); return 0; }
}  // namespace fd

struct fd_kernel_s {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    const void *host_fn = nullptr;   // builtin (compiled into this library)
    size_t max_lds_set = 0;
};
struct fd_event_s { hipEvent_t ev; };
struct fd_graph_s { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; hipStream_t stream = nullptr;
                    hipStream_t last = nullptr; bool last_is_default = false; };      // (the stream of the last launch: what fd_graph_sync waits for)

namespace {
// zero-fill of a small 8-byte-aligned range in ONE launch: hipMemsetAsync splits a range whose size is not a multiple of its block
// into two fill kernels (bulk + tail: 2 x 2.5 us of a 22 us step of config C1, where a residual vector is 33 800 bytes)
__global__ void zero_words_k(unsigned long long *__restrict__ p, size_t nwords) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) p[i] = 0ull;
}
}  // namespace

extern "C" {

int fd_version(void) { return 100; }
const char *fd_last_error(void) { return fd::g_err.c_str(); }

int fd_device_count(int *n) { FD_HIP(hipGetDeviceCount(n)); return 0; }
int fd_set_device(int device) { FD_HIP(hipSetDevice(device)); return 0; }

int fd_device_info(int device, char *name, size_t name_len, int *cus, size_t *hbm, int *lds) {
    hipDeviceProp_t p;
    if (device < 0) FD_HIP(hipGetDevice(&device));            // (the calling thread's current device)
    FD_HIP(hipGetDeviceProperties(&p, device));
    if (name && name_len) { std::strncpy(name, p.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (cus) *cus = p.multiProcessorCount;
    if (hbm) *hbm = p.totalGlobalMem;
    if (lds) *lds = (int)p.sharedMemPerBlock;
    return 0;
}

int fd_malloc(void **ptr, size_t bytes) {
    if (fd::g_capturing) FD_FAIL("fd_malloc inside fd_graph_begin .. fd_graph_end: a captured step may only enqueue work -- run it once before the capture so that its plans and buffers exist");
    FD_HIP(hipMalloc(ptr, bytes ? bytes : 8));
    return 0;
}
int fd_free(void *ptr) {
    FD_HIP(fd::release(ptr));
    return 0;
}
int fd_memset(void *p, int b, size_t n, fd_stream_t s) {
    if (!n) return 0;
    if (b == 0 && n <= (size_t)(4u << 20) && n % 8 == 0 && (reinterpret_cast<uintptr_t>(p) & 7u) == 0) {
        const size_t nw = n / 8;
        const unsigned grid = (unsigned)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048);
        hipLaunchKernelGGL(zero_words_k, dim3(grid), dim3(256), 0, fd::st(s), static_cast<unsigned long long *>(p), nw);
        FD_CHECK_LAUNCH();
        return 0;
    }
    FD_HIP(hipMemsetAsync(p, b, n, fd::st(s)));
    return 0;
}
int fd_memcpy_h2d(void *d, const void *s_, size_t n, fd_stream_t s) { if (n) FD_HIP(hipMemcpyAsync(d, s_, n, hipMemcpyHostToDevice, fd::st(s))); return 0; }
int fd_memcpy_d2h(void *d, const void *s_, size_t n, fd_stream_t s) {
    if (n) { FD_HIP(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToHost, fd::st(s))); FD_HIP(hipStreamSynchronize(fd::st(s))); }
    return 0;
}
int fd_memcpy_d2d(void *d, const void *s_, size_t n, fd_stream_t s) { if (n) FD_HIP(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, fd::st(s))); return 0; }
int fd_stream_create(fd_stream_t *s) { hipStream_t h; FD_HIP(hipStreamCreateWithFlags(&h, hipStreamNonBlocking)); *s = h; return 0; }
int fd_stream_destroy(fd_stream_t s) {
    if (!s) return 0;
    if (fd::g_capturing) { fd::g_deferred_streams.push_back(reinterpret_cast<hipStream_t>(s)); return 0; }
    FD_HIP(hipStreamDestroy(reinterpret_cast<hipStream_t>(s)));
    return 0;
}
int fd_stream_sync(fd_stream_t s) { FD_HIP(hipStreamSynchronize(fd::st(s))); return 0; }
int fd_device_sync(void) { FD_HIP(hipDeviceSynchronize()); return 0; }
// The stream every call with a NULL stream argument uses from now on (NULL: back to the HIP null stream).  Two independent
// parloops -- the residual and the Jacobian of one Newton step -- are put on two streams this way and share the device.
int fd_stream_set_default(fd_stream_t s) { fd::g_default_stream = reinterpret_cast<hipStream_t>(s); return 0; }
int fd_stream_get_default(fd_stream_t *s) { if (!s) FD_FAIL("fd_stream_get_default: null result"); *s = reinterpret_cast<fd_stream_t>(fd::g_default_stream); return 0; }
int fd_stream_wait_event(fd_stream_t s, fd_event_t e) { if (!e) FD_FAIL("fd_stream_wait_event: null event"); FD_HIP(hipStreamWaitEvent(fd::st(s), e->ev, 0)); return 0; }

int fd_event_create(fd_event_t *e) { auto *p = new fd_event_s; hipError_t r = hipEventCreate(&p->ev); if (r != hipSuccess) { delete p; FD_HIP(r); } *e = p; return 0; }
int fd_event_destroy(fd_event_t e) { if (e) { FD_HIP(hipEventDestroy(e->ev)); delete e; } return 0; }
int fd_event_record(fd_event_t e, fd_stream_t s) { FD_HIP(hipEventRecord(e->ev, fd::st(s))); return 0; }
int fd_event_sync(fd_event_t e) { FD_HIP(hipEventSynchronize(e->ev)); return 0; }
int fd_event_elapsed_ms(fd_event_t a, fd_event_t b, float *ms) { FD_HIP(hipEventElapsedTime(ms, a->ev, b->ev)); return 0; }

// ---- hipGraph capture of a launch-bound assembly step (small meshes: Python + launch overhead dominates,
// SURVEY.md 7 hard part (f)).  Between begin and end every libfdhip call that takes a NULL stream records
// into the graph instead of executing.
int fd_graph_begin(fd_graph_t *out) {
    if (fd::g_capturing) FD_FAIL("fd_graph_begin: a capture is already open (captures do not nest)");
    auto *g = new fd_graph_s;
    hipError_t r = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (r != hipSuccess) { delete g; FD_HIP(r); }
    r = hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal);
    if (r != hipSuccess) { (void)hipStreamDestroy(g->stream); delete g; FD_HIP(r); }
    fd::g_saved_stream = fd::g_default_stream;
    fd::g_default_stream = g->stream;
    fd::g_capturing = true;
    *out = g;
    return 0;
}

int fd_graph_end(fd_graph_t g) {
    fd::g_default_stream = fd::g_saved_stream;
    fd::g_saved_stream = nullptr;
    fd::g_capturing = false;
    if (!g) FD_FAIL("fd_graph_end: null graph");
    const hipError_t ended = hipStreamEndCapture(g->stream, &g->graph);
    for (void *p : fd::g_deferred_free) (void)hipFree(p);           // (releases parked during the capture)
    fd::g_deferred_free.clear();
    for (fd_kernel_t k : fd::g_deferred_kernels) (void)fd_kernel_free(k);
    fd::g_deferred_kernels.clear();
    for (fd_graph_t d : fd::g_deferred_graphs) (void)fd_graph_free(d);
    fd::g_deferred_graphs.clear();
    for (hipStream_t q : fd::g_deferred_streams) (void)hipStreamDestroy(q);
    fd::g_deferred_streams.clear();
    FD_HIP(ended);
    FD_HIP(hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0));
    return 0;
}

int fd_graph_launch(fd_graph_t g, fd_stream_t s) {
    if (!g || !g->exec) FD_FAIL("fd_graph_launch: graph not instantiated");
    g->last = s ? fd::st(s) : g->stream;
    g->last_is_default = false;
    FD_HIP(hipGraphLaunch(g->exec, g->last));
    return 0;
}

int fd_graph_launch_default(fd_graph_t g) {
    if (!g || !g->exec) FD_FAIL("fd_graph_launch_default: graph not instantiated");
    if (fd::g_capturing) FD_FAIL("fd_graph_launch_default: a capture is open");
    g->last = fd::default_stream();
    g->last_is_default = true;
    FD_HIP(hipGraphLaunch(g->exec, g->last));
    return 0;
}

int fd_graph_sync(fd_graph_t g) {
    if (!g) FD_FAIL("fd_graph_sync: null graph");
    FD_HIP(hipStreamSynchronize(g->last_is_default || g->last ? g->last : g->stream));
    return 0;
}

int fd_graph_free(fd_graph_t g) {
    if (!g) return 0;
    if (fd::g_capturing) { fd::g_deferred_graphs.push_back(g); return 0; }
    if (g->exec) FD_HIP(hipGraphExecDestroy(g->exec));
    if (g->graph) FD_HIP(hipGraphDestroy(g->graph));
    if (g->stream) FD_HIP(hipStreamDestroy(g->stream));
    delete g;
    return 0;
}

int fd_kernel_load(const char *path, const char *symbol, fd_kernel_t *out) {
    auto *k = new fd_kernel_s;
    hipError_t r = hipModuleLoad(&k->mod, path);
    if (r != hipSuccess) { delete k; fd::set_error(std::string("hipModuleLoad(") + path + "): " + hipGetErrorString(r)); return (int)r; }
    r = hipModuleGetFunction(&k->fn, k->mod, symbol);
    if (r != hipSuccess) { (void)hipModuleUnload(k->mod); delete k; fd::set_error(std::string("symbol ") + symbol + " not in " + path + ": " + hipGetErrorString(r)); return (int)r; }
    *out = k;
    return 0;
}

// JIT from wrapper source inside the library (for hosts that do not want to shell out themselves): the counterpart of
// compilation.load() -> make_so() (pyop2/compilation.py:424-455, 527-611): hash (source, flags, compiler) -> cached code
// object under cache_dir, else hipcc --genco into a temporary name + rename (concurrent ranks race benignly), then load.
static uint64_t fnv1a(const std::string &s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

// contents of the device headers a wrapper includes + the compiler's identity (path, size, mtime): part of the cache key, so a
// header change or a hipcc upgrade never reuses a stale code object (compilation.compile_hip keys its cache the same way)
static uint64_t toolchain_hash(const std::string &inc, const char *hipcc) {
    uint64_t h = fnv1a(hipcc);
    struct stat st;
    if (stat(hipcc, &st) == 0) h = fnv1a(std::to_string((long long)st.st_size) + ":" + std::to_string((long long)st.st_mtime), h);
    for (const char *name : {"fd_wrapper.h", "fd_tensor.h", "fd_callables.h"}) {
        const std::string path = inc + "/" + name;
        if (FILE *f = fopen(path.c_str(), "rb")) {
            char buf[4096];
            size_t n;
            while ((n = fread(buf, 1, sizeof buf, f)) > 0) h = fnv1a(std::string(buf, n), h);
            fclose(f);
        }
    }
    return h;
}

// run `argv` (no shell: nothing in the flags is ever interpreted) with stdout + stderr in `log`; returns the exit status
static int run_logged(const std::vector<std::string> &argv, const std::string &log) {
    std::vector<char *> av;
    for (const auto &a : argv) av.push_back(const_cast<char *>(a.c_str()));
    av.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid = 0;
    const int rc = posix_spawnp(&pid, av[0], &fa, nullptr, av.data(), environ);      // (searches PATH: FDHIP_HIPCC=hipcc works)
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) {
        if (FILE *f = fopen(log.c_str(), "w")) { fprintf(f, "cannot start %s: %s\n", av[0], strerror(rc)); fclose(f); }
        return -1;
    }
    int status = 0;
    if (waitpid(pid, &status, 0) < 0) return -1;
    return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

int fd_kernel_create(const char *wrapper_src, const char *symbol, const char *cache_dir, const char *extra_flags, fd_kernel_t *out) {
    if (!wrapper_src || !symbol || !cache_dir || !out) FD_FAIL("fd_kernel_create: bad arguments");
    const char *hipcc = getenv("FDHIP_HIPCC");
    if (!hipcc) hipcc = "/opt/rocm/bin/hipcc";
    const char *arch = getenv("FDHIP_ARCH");
    if (!arch) arch = "gfx950";
    // the device headers (fd_wrapper.h, fd_tensor.h) live next to this library: <dir of libfdhip.so>/csrc
    Dl_info info;
    std::string inc;
    if (dladdr((const void *)&fd_kernel_create, &info) && info.dli_fname) {
        std::string lib(info.dli_fname);
        size_t p = lib.find_last_of('/');
        inc = (p == std::string::npos ? std::string(".") : lib.substr(0, p)) + "/csrc";
    }
    std::vector<std::string> argv = {hipcc, std::string("--offload-arch=") + arch, "-O3", "-std=c++17", "--genco", "-munsafe-fp-atomics",
                                     "-fno-math-errno", "-fno-signed-zeros", "-fno-honor-nans", "-fno-honor-infinities",
                                     "-fassociative-math", "-fno-trapping-math", "-ffp-contract=fast"};
    if (extra_flags) {                                   // whitespace-separated extra flags, passed as separate arguments
        std::istringstream ss(extra_flags);
        for (std::string tok; ss >> tok;) argv.push_back(tok);
    }
    std::string flags;
    for (size_t i = 1; i < argv.size(); ++i) flags += argv[i] + " ";
    char key[32];
    snprintf(key, sizeof key, "%016llx", (unsigned long long)fnv1a(flags, fnv1a(wrapper_src, toolchain_hash(inc, hipcc))));
    const std::string base = std::string(cache_dir) + "/" + symbol + "_c" + key;
    const std::string obj = base + ".hsaco";
    if (access(obj.c_str(), R_OK) != 0) {
        const std::string tmp = base + "." + std::to_string((long)getpid()) + ".tmp";
        const std::string src = tmp + ".hip";
        FILE *f = fopen(src.c_str(), "w");
        if (!f) FD_FAIL("fd_kernel_create: cannot write " + src);
        fputs(wrapper_src, f);
        fclose(f);
        const std::string log = tmp + ".log";
        argv.push_back("-I" + inc);
        argv.push_back("-o"); argv.push_back(tmp + ".hsaco");
        argv.push_back(src);
        const int rc = run_logged(argv, log);
        if (rc != 0) {
            std::string msg = "fd_kernel_create: hipcc failed (" + std::string(hipcc) + " " + flags + "... exit " + std::to_string(rc) + ")";
            if (FILE *lf = fopen(log.c_str(), "r")) { char buf[2048]; size_t n = fread(buf, 1, sizeof buf - 1, lf); buf[n] = 0; fclose(lf); msg += std::string("\n") + buf; }
            unlink(src.c_str()); unlink(log.c_str()); unlink((tmp + ".hsaco").c_str());
            FD_FAIL(msg);
        }
        unlink(src.c_str()); unlink(log.c_str());
        if (rename((tmp + ".hsaco").c_str(), obj.c_str()) != 0) FD_FAIL("fd_kernel_create: cannot move the code object into the cache");
    }
    return fd_kernel_load(obj.c_str(), symbol, out);
}

int fd_kernel_builtin(const char *symbol, fd_kernel_t *out) {
    for (auto &b : fd::registry())
        if (std::strcmp(b.name, symbol) == 0) { auto *k = new fd_kernel_s; k->host_fn = b.fn; *out = k; return 0; }
    FD_FAIL(std::string("no builtin wrapper kernel named ") + symbol);
}

int fd_kernel_free(fd_kernel_t k) {
    if (!k) return 0;
    if (fd::g_capturing) { fd::g_deferred_kernels.push_back(k); return 0; }
    if (k->mod) FD_HIP(hipModuleUnload(k->mod));
    delete k;
    return 0;
}

int fd_kernel_launch(fd_kernel_t k, int32_t start, int32_t end, const void *const *args, int nargs,
                     int block_threads, int ents_per_block, int nblocks, size_t lds_bytes, fd_stream_t s) {
    if (!k) FD_FAIL("fd_kernel_launch: null kernel");
    if (end <= start) return 0;     // empty iteration range: nothing to do (builder.py:734-741)
    if (block_threads <= 0 || block_threads > 1024) FD_FAIL("fd_kernel_launch: bad block size");
    if (nblocks <= 0) {
        if (ents_per_block <= 0) FD_FAIL("fd_kernel_launch: need ents_per_block or nblocks");
        int64_t n = (int64_t)end - start;
        nblocks = (int)((n + ents_per_block - 1) / ents_per_block);
    }
    void *params[64];
    uint64_t vals[64];
    if (nargs > 62) FD_FAIL("fd_kernel_launch: too many arguments");
    params[0] = &start; params[1] = &end;
    for (int i = 0; i < nargs; ++i) { vals[i] = (uint64_t)(uintptr_t)args[i]; params[2 + i] = &vals[i]; }
    if (lds_bytes > 64 * 1024 && lds_bytes > k->max_lds_set) {
        if (k->host_fn) FD_HIP(hipFuncSetAttribute(k->host_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        else FD_HIP(hipFuncSetAttribute((const void *)k->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        k->max_lds_set = lds_bytes;
    }
    if (k->host_fn) {
        FD_HIP(hipLaunchKernel(k->host_fn, dim3(nblocks), dim3(block_threads), params, lds_bytes, fd::st(s)));
    } else {
        FD_HIP(hipModuleLaunchKernel(k->fn, nblocks, 1, 1, block_threads, 1, 1, (unsigned)lds_bytes, fd::st(s), params, nullptr));
    }
    return 0;
}

// ---- tracing ranges (pyop2/profiling.py timed_region, pyop2/parloop.py:219-232): roctx markers around the launches of a parloop.
// libroctx64 is bound at run time -- a process that never traces never loads it.
namespace {
typedef int (*roctx_push_t)(const char *);
typedef int (*roctx_pop_t)();
roctx_push_t roctx_push = nullptr;
roctx_pop_t roctx_pop = nullptr;
int roctx_state = 0;      // 0 = not tried, 1 = bound, -1 = unavailable
bool roctx_bind() {
    if (roctx_state == 0) {
        roctx_state = -1;
        for (const char *lib : {"libroctx64.so.4", "libroctx64.so", "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so"}) {
            void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            roctx_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
            roctx_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
            if (roctx_push && roctx_pop) { roctx_state = 1; break; }
        }
    }
    return roctx_state == 1;
}
}  // namespace

int fd_trace_available(void) { return roctx_bind() ? 1 : 0; }

int fd_trace_range_push(const char *name) {
    if (!name) FD_FAIL("fd_trace_range_push: no name");
    if (!roctx_bind()) FD_FAIL("fd_trace_range_push: libroctx64 is not available");
    roctx_push(name);
    return 0;
}

int fd_trace_range_pop(void) {
    if (!roctx_bind()) FD_FAIL("fd_trace_range_pop: libroctx64 is not available");
    roctx_pop();
    return 0;
}

}  // extern "C"
