// Runtime plumbing of libaozora_hip.so: device query, hipGraph capture/replay of a launch
// sequence (the static SDXL train-step program), pinned host memory for Raven/Titan state, and
// HIP events for timing on the stream the kernels actually run on.
#include "az_common.h"
#include "aozora_hip.h"
#include <string.h>

namespace {
// one wave that idles for `ticks` of the 100 MHz wall clock: the probe the executor uses to find out whether two HIP streams
// really run side by side (streams share a handful of hardware queues; two that land on one pipe serialise)
__global__ void spin_kernel(long ticks) {
  const long t0 = (long)wall_clock64();
  while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

extern "C" {

int az_version(void) { return 100; }

int az_spin(long microseconds, void* stream) {
  if (microseconds < 0 || microseconds > 100000) return AZ_ERR_ARG(90);
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, microseconds * 100);
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
