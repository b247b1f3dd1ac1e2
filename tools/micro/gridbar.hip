// micro-benchmark: cost of a grid barrier among a team of workgroups (every `stride`-th workgroup of the launch; workgroups are
// dealt round-robin over the 8 XCDs, so stride 8 = one XCD), with agent-scope release / acquire around a single-counter barrier.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/micro/gridbar.hip -o /tmp/gridbar && /tmp/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(512) k_bar(unsigned *ctr, int nbar, int stride, long long *out, double *buf, int *err) {
  if (blockIdx.x % stride) return;
  const int team = gridDim.x / stride, me = blockIdx.x / stride;
  const int nt = blockDim.x, gid = me * nt + threadIdx.x, n = team * nt;
  unsigned target = 0;
  int bad = 0;
  const long long t0 = wall_clock64();
  for (int b = 1; b <= nbar; b++) {
    double *cur = buf + (size_t)(b & 1) * n;   // double-buffered: a fast workgroup's next write does not race a slow one's read
    cur[gid] = (double)b;
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope by default in HIP
      atomicAdd(ctr, 1u);
      target += team;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    const double v = cur[(gid + nt * 3 + 17) % n];  // a value written by another workgroup in this phase
    if (v != (double)b) bad++;
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[me] = t1 - t0; }
  if (bad) atomicAdd(err, bad);
}

int main() {
  unsigned *ctr; long long *out; double *buf; int *err;
  if (hipMalloc(&ctr, 4) || hipMalloc(&out, 256 * 8) || hipMalloc(&buf, 2 * 256 * 512 * 8) || hipMalloc(&err, 4)) return 1;
  const int nbar = 200;
  for (int nwg : {256})   // one 512-thread workgroup per CU: always co-resident on an idle MI355X (buffers below are sized for 256)
    for (int stride : {1, 2, 4, 8, 16, 32}) {
      for (int rep = 0; rep < 2; rep++) {
        hipMemset(ctr, 0, 4); hipMemset(err, 0, 4); hipMemset(buf, 0, 2 * 256 * 512 * 8);
        hipLaunchKernelGGL(k_bar, dim3(nwg), dim3(512), 0, 0, ctr, nbar, stride, out, buf, err);
        hipDeviceSynchronize();
        std::vector<long long> h(256); int e = 0;
        hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost); hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        if (rep) printf("grid %d stride %d team %d: %.2f us per phase (write + barrier + read), errors %d\n", nwg, stride, nwg / stride, h[0] / 100.0 / nbar, e);
      }
    }
  return 0;
}
