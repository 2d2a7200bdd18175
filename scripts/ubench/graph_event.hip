// Which way of putting a timestamped event INSIDE a captured hipGraph works on this ROCm?  (measurement tooling)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(float *p, int n) { float x = p[threadIdx.x]; for (int i = 0; i < n; i++) x = x * 1.0001f + 0.5f; p[threadIdx.x] = x; }
#define CK(x) do { hipError_t e_ = (x); printf("%-60s -> %d (%s)\n", #x, (int)e_, hipGetErrorName(e_)); } while (0)
int main() {
  hipStream_t st; hipStreamCreate(&st);
  float *d; hipMalloc(&d, 1024);
  hipEvent_t a, b, c, e;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c)); CK(hipEventCreate(&e));
  for (int mode = 0; mode < 2; mode++) {
    printf("---- mode %d (%s)\n", mode, mode == 0 ? "hipEventRecordWithFlags external" : "explicit event-record nodes");
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    hipEvent_t s0 = mode == 0 ? a : c, s1 = mode == 0 ? b : e;
    auto mark = [&](hipEvent_t ev) {
      if (mode == 0) { CK(hipEventRecordWithFlags(ev, st, hipEventRecordExternal)); return; }
      hipStreamCaptureStatus cs; unsigned long long id; hipGraph_t cg; const hipGraphNode_t *deps; size_t nd;
      CK(hipStreamGetCaptureInfo_v2(st, &cs, &id, &cg, &deps, &nd));
      hipGraphNode_t node;
      CK(hipGraphAddEventRecordNode(&node, cg, deps, nd, ev));
      CK(hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies));
    };
    mark(s0);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 2000000);
    mark(s1);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; r++) {
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      float ms = -1; CK(hipEventElapsedTime(&ms, s0, s1));
      printf("replay %d: %.3f ms\n", r, ms);
    }
  }
  return 0;
}
