// Error reporting, dispatch, device info and the program (launch-list / hipGraph) executor.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "ck_internal.h"

namespace {
thread_local char g_err[512] = "";
thread_local ck_program* g_recording = nullptr;
}  // namespace

struct ck_program {
  std::vector<ck::Launch> ops;
  const void* inputs[ck::kProgramInputs] = {nullptr, nullptr, nullptr, nullptr};  // ck_program_set_input
  bool finished = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

namespace ck {

int fail(ck_status st, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return static_cast<int>(st);
}

int dispatch(Launch fn, void* stream) {
  if (g_recording != nullptr) {
    g_recording->ops.push_back(std::move(fn));
    return CK_OK;
  }
  hipError_t e = fn(static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(CK_ERR_HIP, "HIP launch failed: %s", hipGetErrorString(e));
  return CK_OK;
}

}  // namespace ck

namespace ck {
const void* const* program_input_slot(int index) {
  if (g_recording == nullptr || index < 0 || index >= kProgramInputs) return nullptr;
  return &g_recording->inputs[index];  // (a program lives on the heap until ck_program_destroy: the address is stable)
}
}  // namespace ck

namespace {
thread_local ck::Workspace g_workspace{nullptr, 0};
}
namespace ck {
Workspace workspace() { return g_workspace; }
int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
}  // namespace ck

extern "C" {

const char* ck_last_error(void) { return g_err; }

int ck_abi_version(void) { return 48; }

int ck_device_info(int device, int64_t out[4]) {
  if (out == nullptr) return ck::fail(CK_ERR_INVALID, "ck_device_info: out is null");
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) return ck::fail(CK_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  out[0] = p.multiProcessorCount;
  out[1] = static_cast<int64_t>(p.maxSharedMemoryPerMultiProcessor);
  out[2] = p.warpSize;
  out[3] = strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
  return CK_OK;
}

int ck_set_workspace(void* ptr, int64_t bytes) {
  if ((ptr == nullptr) != (bytes <= 0)) return ck::fail(CK_ERR_INVALID, "ck_set_workspace: pointer and size disagree");
  if (ptr != nullptr && !ck::aligned16(ptr)) return ck::fail(CK_ERR_INVALID, "ck_set_workspace: not 16-byte aligned");
  g_workspace = {ptr, bytes > 0 ? bytes : 0};
  return CK_OK;
}

int ck_program_begin(ck_program** out) {
  if (out == nullptr) return ck::fail(CK_ERR_INVALID, "ck_program_begin: out is null");
  if (g_recording != nullptr) return ck::fail(CK_ERR_STATE, "a program is already being recorded on this thread");
  *out = new ck_program();
  g_recording = *out;
  return CK_OK;
}

int ck_program_end(ck_program* prog) {
  if (prog == nullptr || g_recording != prog) return ck::fail(CK_ERR_STATE, "ck_program_end: not the recording program");
  g_recording = nullptr;
  prog->finished = true;
  return CK_OK;
}

int ck_program_set_input(ck_program* prog, int index, const void* ptr) {
  if (prog == nullptr || index < 0 || index >= ck::kProgramInputs)
    return ck::fail(CK_ERR_INVALID, "ck_program_set_input: null program or index %d outside [0, %d)", index, ck::kProgramInputs);
  prog->inputs[index] = ptr;
  return CK_OK;
}

int ck_program_num_ops(const ck_program* prog) {
  return prog == nullptr ? -1 : static_cast<int>(prog->ops.size());
}

int ck_program_launch(ck_program* prog, int use_graph, void* stream_) {
  if (prog == nullptr || !prog->finished) return ck::fail(CK_ERR_STATE, "ck_program_launch: program not finished");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!use_graph) {
    for (auto& op : prog->ops) {
      hipError_t e = op(stream);
      if (e != hipSuccess) return ck::fail(CK_ERR_HIP, "program op failed: %s", hipGetErrorString(e));
    }
    return CK_OK;
  }
  if (prog->exec == nullptr) {
    hipError_t e = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return ck::fail(CK_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    hipError_t op_err = hipSuccess;
    for (auto& op : prog->ops) {
      op_err = op(stream);
      if (op_err != hipSuccess) break;
    }
    e = hipStreamEndCapture(stream, &prog->graph);
    if (op_err != hipSuccess) return ck::fail(CK_ERR_HIP, "program op failed in capture: %s", hipGetErrorString(op_err));
    if (e != hipSuccess) return ck::fail(CK_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&prog->exec, prog->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) return ck::fail(CK_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  hipError_t e = hipGraphLaunch(prog->exec, stream);
  if (e != hipSuccess) return ck::fail(CK_ERR_HIP, "hipGraphLaunch: %s", hipGetErrorString(e));
  return CK_OK;
}

int ck_program_destroy(ck_program* prog) {
  if (prog == nullptr) return CK_OK;
  if (g_recording == prog) g_recording = nullptr;
  if (prog->exec != nullptr) (void)hipGraphExecDestroy(prog->exec);
  if (prog->graph != nullptr) (void)hipGraphDestroy(prog->graph);
  delete prog;
  return CK_OK;
}

}  // extern "C"
