// Micro-benchmark: cost of wave-level returning atomicAdd work counters on MI355X.
// same-address counter vs 64 counters on separate 128-B lines.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void grab(unsigned* ctr, int nq, int stride, unsigned total, unsigned* sink) {
  unsigned got = 0, acc = 0;
  int q = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % nq;
  for (;;) {
    unsigned t = 0;
    if ((threadIdx.x & 63) == 0) t = atomicAdd(&ctr[q * stride], 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= total / nq) break;
    acc += t; ++got;
  }
  if ((threadIdx.x & 63) == 0) atomicAdd(sink, got + (acc & 1));
}
int main() {
  unsigned *ctr, *sink; hipMalloc(&ctr, 64 * 128); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nq : {1, 8, 64}) for (unsigned total : {32768u, 262144u}) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(ctr, 0, 64 * 128); hipMemset(sink, 0, 4);
      hipEventRecord(e0); grab<<<2048, 256>>>(ctr, nq, 32, total, sink); hipEventRecord(e1);
      hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("queues=%2d grabs=%7u: %.3f ms  (%.1f ns per grab)\n", nq, total, best, best * 1e6 / total);
  }
  return 0;
}
