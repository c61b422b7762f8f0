// tools/stream_sync_bench.hip — what does it cost to order a long kernel on one stream behind a short kernel on another, per step?
// Models a steady-state submit: pre(i+1) [short kernel, stream P] may start when main(i-1) is done and must finish before main(i+1)
// [long kernel, stream M]. Variants: HIP events (default flags / hipEventDisableSystemFence), stream memory operations
// (hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory), and no cross-stream ordering at all (lower bound).
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_sync_bench tools/stream_sync_bench.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(unsigned long long ticks, unsigned* sink) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < ticks) {}
  if (ticks == 1) *sink = 1;
}

int main() {
  int can = 0;
  CHECK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t M, P;
  CHECK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&P, hipStreamNonBlocking));
  unsigned* sink;
  CHECK(hipMalloc((void**)&sink, 4));
  const unsigned long long LONG_T = 800000, SHORT_T = 50000;  // ~240 us / ~15 us (the counter runs at a few GHz here)
  const int steps = 200;
  // calibrate: one long kernel alone
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& body) {
    body(10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, M));
    body(steps);
    CHECK(hipEventRecord(e1, M));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-56s %.2f us per step\n", name, ms * 1e3 / steps);
  };
  timeit("long kernels back to back on one stream", [&](int n) {
    for (int i = 0; i < n; ++i) spin_kernel<<<256, 64, 0, M>>>(LONG_T, sink);
  });
  for (int fence = 0; fence < 2; ++fence) {
    const unsigned fl = hipEventDisableTiming | (fence ? hipEventDisableSystemFence : 0);
    hipEvent_t pre_done[2], main_done[2];
    for (int b = 0; b < 2; ++b) {
      CHECK(hipEventCreateWithFlags(&pre_done[b], fl));
      CHECK(hipEventCreateWithFlags(&main_done[b], fl));
    }
    bool valid[2] = {false, false};
    timeit(fence ? "events, hipEventDisableSystemFence" : "events, default fence", [&](int n) {
      for (int i = 0; i < n; ++i) {
        const int b = i & 1;
        if (valid[b]) CHECK(hipStreamWaitEvent(P, main_done[b], 0));
        spin_kernel<<<64, 64, 0, P>>>(SHORT_T, sink);
        CHECK(hipEventRecord(pre_done[b], P));
        CHECK(hipStreamWaitEvent(M, pre_done[b], 0));
        spin_kernel<<<256, 64, 0, M>>>(LONG_T, sink);
        CHECK(hipEventRecord(main_done[b], M));
        valid[b] = true;
      }
    });
  }
  if (can) {
    unsigned *sig0, *sig1;  // (signal memory comes in 8-byte allocations)
    CHECK(hipExtMallocWithFlags((void**)&sig0, 8, hipMallocSignalMemory));
    CHECK(hipExtMallocWithFlags((void**)&sig1, 8, hipMallocSignalMemory));
    CHECK(hipMemset(sig0, 0, 8));
    CHECK(hipMemset(sig1, 0, 8));
    unsigned step_no = 0;
    timeit("stream memory ops (write/wait value on signal memory)", [&](int n) {
      for (int i = 0; i < n; ++i) {
        ++step_no;
        // pre(step) may start when main(step - 2) is done: sig[1] >= step - 2
        if (step_no > 2) CHECK(hipStreamWaitValue32(P, sig1, step_no - 2, hipStreamWaitValueGte, 0xFFFFFFFFu));
        spin_kernel<<<64, 64, 0, P>>>(SHORT_T, sink);
        CHECK(hipStreamWriteValue32(P, sig0, step_no, 0));
        CHECK(hipStreamWaitValue32(M, sig0, step_no, hipStreamWaitValueGte, 0xFFFFFFFFu));
        spin_kernel<<<256, 64, 0, M>>>(LONG_T, sink);
        CHECK(hipStreamWriteValue32(M, sig1, step_no, 0));
      }
    });
  }
  {
    // the same step as a captured graph: pre(i+1) forked from main(i-1), joined before main(i+1); two steps per graph so that the
    // steady-state dependency pattern is inside it (launch k+1 still serialises behind launch k on stream M)
    hipGraph_t g;
    hipGraphExec_t ge;
    hipEvent_t f0, f1, j0, j1;
    CHECK(hipEventCreateWithFlags(&f0, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&f1, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&j0, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&j1, hipEventDisableTiming));
    CHECK(hipStreamBeginCapture(M, hipStreamCaptureModeGlobal));
    CHECK(hipEventRecord(f0, M));
    CHECK(hipStreamWaitEvent(P, f0, 0));
    spin_kernel<<<64, 64, 0, P>>>(SHORT_T, sink);   // pre(a)
    CHECK(hipEventRecord(j0, P));
    spin_kernel<<<64, 64, 0, P>>>(SHORT_T, sink);   // pre(b): may overlap main(a)
    CHECK(hipEventRecord(j1, P));
    CHECK(hipStreamWaitEvent(M, j0, 0));
    spin_kernel<<<256, 64, 0, M>>>(LONG_T, sink);   // main(a)
    CHECK(hipStreamWaitEvent(M, j1, 0));
    spin_kernel<<<256, 64, 0, M>>>(LONG_T, sink);   // main(b)
    CHECK(hipStreamEndCapture(M, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    timeit("captured graph, two steps per launch", [&](int n) {
      for (int i = 0; i < n; i += 2) CHECK(hipGraphLaunch(ge, M));
    });
  }
  {
    // round 3: ONE stream. The short kernel of step i+1 is launched behind the long kernel of step i with hipExtAnyOrderLaunch (no barrier
    // bit on its dispatch packet: it need not wait for the long kernel), the next long kernel is an ordinary launch (waits for everything
    // before it). No events, no second queue.
    timeit("one stream, short kernel launched any-order", [&](int n) {
      for (int i = 0; i < n; ++i) {
        hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, M, nullptr, nullptr, hipExtAnyOrderLaunch, SHORT_T, sink);
        spin_kernel<<<256, 64, 0, M>>>(LONG_T, sink);
      }
    });
    timeit("one stream, short kernel ordinary launch (serial)", [&](int n) {
      for (int i = 0; i < n; ++i) {
        spin_kernel<<<64, 64, 0, M>>>(SHORT_T, sink);
        spin_kernel<<<256, 64, 0, M>>>(LONG_T, sink);
      }
    });
    // events attached to the kernels' own dispatch packets (stopEvent of hipExtLaunchKernelGGL) instead of hipEventRecord
    const unsigned fl = hipEventDisableTiming | hipEventDisableSystemFence;
    hipEvent_t pre_done[2], main_done[2];
    for (int b = 0; b < 2; ++b) {
      CHECK(hipEventCreateWithFlags(&pre_done[b], fl));
      CHECK(hipEventCreateWithFlags(&main_done[b], fl));
    }
    bool valid[2] = {false, false};
    timeit("events as the kernels' stopEvent (fence-free)", [&](int n) {
      for (int i = 0; i < n; ++i) {
        const int b = i & 1;
        if (valid[b]) CHECK(hipStreamWaitEvent(P, main_done[b], 0));
        hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, P, nullptr, pre_done[b], 0, SHORT_T, sink);
        CHECK(hipStreamWaitEvent(M, pre_done[b], 0));
        hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, M, nullptr, main_done[b], 0, LONG_T, sink);
        valid[b] = true;
      }
    });
    // a ring of 4 workspaces: pre(i+1) only waits for main(i-3), so its event is long complete when main(i+1) is reached
    hipEvent_t pd[4], md[4];
    bool v4[4] = {false, false, false, false};
    for (int b = 0; b < 4; ++b) {
      CHECK(hipEventCreateWithFlags(&pd[b], fl));
      CHECK(hipEventCreateWithFlags(&md[b], fl));
    }
    timeit("events, fence-free, ring of 4 (pre runs 3 steps ahead)", [&](int n) {
      for (int i = 0; i < n; ++i) {
        const int b = i & 3;
        if (v4[b]) CHECK(hipStreamWaitEvent(P, md[b], 0));
        spin_kernel<<<64, 64, 0, P>>>(SHORT_T, sink);
        CHECK(hipEventRecord(pd[b], P));
        CHECK(hipStreamWaitEvent(M, pd[b], 0));
        spin_kernel<<<256, 64, 0, M>>>(LONG_T, sink);
        CHECK(hipEventRecord(md[b], M));
        v4[b] = true;
      }
    });
    // only the wait on M (no record on M at all: as if the workspace ring were deep enough never to need one)
    timeit("events, fence-free, only pre_done (no record on M)", [&](int n) {
      for (int i = 0; i < n; ++i) {
        const int b = i & 3;
        spin_kernel<<<64, 64, 0, P>>>(SHORT_T, sink);
        CHECK(hipEventRecord(pd[b], P));
        CHECK(hipStreamWaitEvent(M, pd[b], 0));
        spin_kernel<<<256, 64, 0, M>>>(LONG_T, sink);
      }
    });
  }
  timeit("two streams, no ordering between them (lower bound)", [&](int n) {
    for (int i = 0; i < n; ++i) {
      spin_kernel<<<64, 64, 0, P>>>(SHORT_T, sink);
      spin_kernel<<<256, 64, 0, M>>>(LONG_T, sink);
    }
  });
  return 0;
}
