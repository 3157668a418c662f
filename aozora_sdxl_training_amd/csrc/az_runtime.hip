// Runtime plumbing of libaozora_hip.so: device query, hipGraph capture/replay of a launch
// sequence (the static SDXL train-step program), pinned host memory for Raven/Titan state, and
// HIP events for timing on the stream the kernels actually run on.
#include "az_common.h"
#include "aozora_hip.h"
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>

namespace {
struct OptDef { const char* name; int def; };
const OptDef OPT_DEFS[AZ_OPT_COUNT] = {
  {"TILE_POLICY", 4}, {"BIG_FILL", 5}, {"SPLIT_SLOTS", 384}, {"NOSPLIT_TILES", 256}, {"LDS_EXCLUSIVE", 0}, {"NT_SPLIT_BIG", 3},
  {"NT_SPLIT_MINK", 5120}, {"ATTN_SPLIT_TARGET", 384}, {"LN_RPB", 32}, {"INKERNEL_FINISH", 0}, {"NT_SPLIT2_MINK", 0}, {"GEMM_ABLATE", 0}, {"GEMM8", 1}, {"NT_SPLIT_FWD", 0}, {"ATTN_PIPE", 15}, {"XCD_SPLIT", 1}, {"ATTN_XCD", 15}, {"NORM_STAT_BF16", 1}, {"GN_RPT", 4}, {"GN_RPT_BWD", 32}, {"GN_RPT_APPLY", 32},
};
std::atomic<int> g_opt[AZ_OPT_COUNT];
std::once_flag g_opt_once;
void opt_init() {
  for (int i = 0; i < AZ_OPT_COUNT; ++i) {
    char key[64];
    snprintf(key, sizeof key, "AZ_%s", OPT_DEFS[i].name);
    const char* e = getenv(key);
    g_opt[i].store(e ? atoi(e) : OPT_DEFS[i].def, std::memory_order_relaxed);
  }
}

// one wave that idles for `ticks` of the 100 MHz wall clock: the probe the executor uses to find out whether two HIP streams
// really run side by side (streams share a handful of hardware queues; two that land on one pipe serialise)
__global__ void spin_kernel(long ticks) {
  const long t0 = (long)wall_clock64();
  while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

// A context (az_init): one per device / per owner, with its OWN option table.  While a context is current on a thread
// (az_make_current) the launchers issued from that thread read ITS table; without one they read the process-wide table above.
// Lifetime: the owner (az_init .. az_destroy) holds one reference and so does every thread that has the context current; the
// memory goes with the LAST of them, so a thread that still has a destroyed context current (an owner collected on another
// thread) keeps reading a valid option table until it makes something else current or exits.
struct AzContext { int device; std::atomic<int> refs; std::atomic<int> opt[AZ_OPT_COUNT]; };
static void ctx_release(AzContext* c) {
  if (c && c->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete c;
}
struct AzCurrent {
  AzContext* c = nullptr;
  ~AzCurrent() { ctx_release(c); }             // thread exit drops the thread's reference
  operator AzContext*() const { return c; }
  AzContext* operator->() const { return c; }
  void set(AzContext* n) {
    if (n == c) return;
    if (n) n->refs.fetch_add(1, std::memory_order_relaxed);
    AzContext* old = c; c = n;
    ctx_release(old);
  }
};
static thread_local AzCurrent t_ctx;

int az_opt(int id) {
  std::call_once(g_opt_once, opt_init);
  AzContext* c = t_ctx.c;
  return c ? c->opt[id].load(std::memory_order_relaxed) : g_opt[id].load(std::memory_order_relaxed);
}

// stop event of this thread's launches (az_set_launch_stop_event) and how many launches have carried it
struct AzStopEvent { hipEvent_t ev = nullptr; long launches = 0; };
static AzStopEvent& az_stop_event_slot() {
  static thread_local AzStopEvent s;
  return s;
}
hipEvent_t az_stop_event_of_this_thread() {
  AzStopEvent& s = az_stop_event_slot();
  if (s.ev) ++s.launches;
  return s.ev;
}

extern "C" {

int az_version(void) { return 101; }

// az_set_option / az_get_option address the calling thread's current context when it has one, else the process-wide table
int az_set_option(const char* name, int value) {
  std::call_once(g_opt_once, opt_init);
  if (!name) return AZ_ERR_ARG(91);
  std::atomic<int>* tab = t_ctx.c ? t_ctx.c->opt : g_opt;
  for (int i = 0; i < AZ_OPT_COUNT; ++i)
    if (!strcmp(name, OPT_DEFS[i].name)) { tab[i].store(value, std::memory_order_relaxed); return AZ_OK; }
  return AZ_ERR_ARG(92);
}

int az_get_option(const char* name, int* value) {
  std::call_once(g_opt_once, opt_init);
  if (!name || !value) return AZ_ERR_ARG(91);
  std::atomic<int>* tab = t_ctx.c ? t_ctx.c->opt : g_opt;
  for (int i = 0; i < AZ_OPT_COUNT; ++i)
    if (!strcmp(name, OPT_DEFS[i].name)) { *value = tab[i].load(std::memory_order_relaxed); return AZ_OK; }
  return AZ_ERR_ARG(92);
}

int az_init(int device, void** handle) {
  std::call_once(g_opt_once, opt_init);
  if (!handle || device < 0) return AZ_ERR_ARG(93);
  AzContext* c = new AzContext;
  c->device = device;
  c->refs.store(1, std::memory_order_relaxed);
  for (int i = 0; i < AZ_OPT_COUNT; ++i) c->opt[i].store(g_opt[i].load(std::memory_order_relaxed), std::memory_order_relaxed);
  *handle = c;
  return AZ_OK;
}

int az_make_current(void* handle) { t_ctx.set((AzContext*)handle); return AZ_OK; }

int az_context_device(void* handle, int* device) {
  if (!handle || !device) return AZ_ERR_ARG(93);
  *device = ((AzContext*)handle)->device;
  return AZ_OK;
}

int az_destroy(void* handle) {
  if (!handle) return AZ_ERR_ARG(93);
  if (t_ctx == (AzContext*)handle) t_ctx.set(nullptr);
  ctx_release((AzContext*)handle);             // the owner's reference; other threads that have it current keep it alive
  return AZ_OK;
}

int az_spin(long microseconds, void* stream) {
  if (microseconds < 0 || microseconds > 100000) return AZ_ERR_ARG(90);
  az_launch(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, microseconds * 100);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_device_info(int* out3) {
  int dev = 0;
  AZ_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  AZ_HIP(hipGetDeviceProperties(&prop, dev));
  out3[0] = prop.multiProcessorCount;
  out3[1] = strstr(prop.gcnArchName, "gfx950") != nullptr ? 1 : 0;
  out3[2] = (int)prop.maxSharedMemoryPerMultiProcessor;
  return AZ_OK;
}

int az_graph_begin(void* stream) {
  AZ_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return AZ_OK;
}

int az_graph_end(void* stream, void** graph_exec_out) {
  hipGraph_t graph = nullptr;
  AZ_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return -(int)e;
  *graph_exec_out = (void*)exec;
  return AZ_OK;
}

int az_graph_launch(void* graph_exec, void* stream) {
  AZ_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return AZ_OK;
}

int az_graph_destroy(void* graph_exec) {
  AZ_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return AZ_OK;
}

int az_host_alloc(void** ptr_host, long bytes) {
  AZ_HIP(hipHostMalloc(ptr_host, (size_t)bytes, hipHostMallocDefault));
  return AZ_OK;
}

int az_host_free(void* ptr_host) {
  AZ_HIP(hipHostFree(ptr_host));
  return AZ_OK;
}

int az_event_create(void** ev) {
  hipEvent_t e;
  AZ_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return AZ_OK;
}
// Fork / join events of the two-stream executor: never timed, never read by the host, and ordering work of ONE device only, so the
// record needs no system-scope fence (tools/event_cost.cpp: a record between two kernels costs their stream 2.8-3.0 us with the
// default flags -- torch.cuda.Event's -- and 1.2-1.4 us without the fence; a fork with its wait on the other stream 5.3 vs 3.0 us)
int az_event_create_fork(void** ev) {
  hipEvent_t e;
  AZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  *ev = (void*)e;
  return AZ_OK;
}
int az_set_launch_stop_event(void* ev) {
  AzStopEvent& s = az_stop_event_slot();
  // clearing an event that NO launch carried would leave it unrecorded: a later wait on it would not wait -- fail loudly instead
  const bool unused = s.ev != nullptr && s.launches == 0;
  s.ev = (hipEvent_t)ev; s.launches = 0;
  return unused ? AZ_ERR_ARG(58) : AZ_OK;
}
int az_stream_wait_event(void* stream, void* ev) { AZ_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0)); return AZ_OK; }
int az_event_record(void* ev, void* stream) { AZ_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return AZ_OK; }
int az_event_sync(void* ev) { AZ_HIP(hipEventSynchronize((hipEvent_t)ev)); return AZ_OK; }
int az_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
  AZ_HIP(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
  return AZ_OK;
}
int az_event_destroy(void* ev) { AZ_HIP(hipEventDestroy((hipEvent_t)ev)); return AZ_OK; }
int az_stream_sync(void* stream) { AZ_HIP(hipStreamSynchronize((hipStream_t)stream)); return AZ_OK; }
int az_memset_async(void* ptr, int value, long bytes, void* stream) {
  AZ_HIP(hipMemsetAsync(ptr, value, (size_t)bytes, (hipStream_t)stream));
  return AZ_OK;
}
int az_memcpy_async(void* dst, const void* src, long bytes, int kind, void* stream) {
  hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : kind == 3 ? hipMemcpyDeviceToDevice : hipMemcpyDefault;
  AZ_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, k, (hipStream_t)stream));
  return AZ_OK;
}

}  // extern "C"
