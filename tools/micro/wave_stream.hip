// wave_stream.hip -- micro-benchmark of the marching kernels' memory pattern (no arithmetic): every wavefront owns a strip of a
// chunk of rows and, per step (= row), touches `ns` streams (layers / fields: `sstride` bytes apart) with one vector-memory
// instruction each: 64 lanes x BPL bytes, contiguous.  Modes: stores, plain loads, LDS-DMA loads, loads + stores.
// Occupancy is set by the dynamic LDS size (waves per CU), the depth by the counted wait after each step.
// build: hipcc --offload-arch=gfx950 -O3 -o wave_stream tools/micro/wave_stream.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args {
  char *base;
  size_t sstride, pitch;   // bytes between streams, between rows
  int nl, nst, steps, nstrips, strip_bytes, mode, cap;
};

template <int BPL>
__device__ __forceinline__ void st(char *p, unsigned off, double v) {
  if constexpr (BPL == 8) asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(v), "s"(p) : "memory");
  else {
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d o; o.x = v; o.y = v;
    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off), "v"(o), "s"(p) : "memory");
  }
}

template <int BPL>
__global__ void __launch_bounds__(64) k_stream(Args a) {
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int w = blockIdx.x;
  const int strip = w % a.nstrips, chunk = w / a.nstrips;
  const unsigned off = (unsigned)(strip * a.strip_bytes + lane * BPL);
  const unsigned offl = (unsigned)(strip * ((a.mode & 4) ? 960 : 480) + lane * ((a.mode & 4) ? 16 : 8));   // loads: 60 owned lanes of 64
  const bool lin = (a.mode & 8) != 0;   // calibration: every wavefront streams one contiguous range (a memset / memcpy)
  size_t linpos = (size_t)w * a.steps * (a.nl + a.nst) * 1024;
  char *row = a.base + (size_t)chunk * a.steps * a.pitch;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) double *)lds);
  double acc = 0.;
  for (int it = 0; it < a.steps; it++, row += a.pitch) {
    // loads: streams 0 .. nl-1
    if (a.mode & 1) {
      for (int s = 0; s < a.nl; s++) {
        const char *p = lin ? a.base + linpos : row + (size_t)s * a.sstride;
        const unsigned off = lin ? lane * ((a.mode & 4) ? 16 : 8) : offl;
        linpos += 1024;
        if (a.mode & 16) {  // LDS-DMA as the passes issue it: lanes 0-31 one 512-byte piece, lanes 32-63 the piece of the NEXT stream (two layer-rows per instruction)
          unsigned keep;
          const unsigned off2 = (unsigned)(strip * 480 + (lane & 31) * 16) ;
          const char *p2 = p + (size_t)(lane >> 5) * a.sstride;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p2 + off2), "s"(0), "s"(lds0 + (unsigned)(s & 7) * 1024u) : "memory");
          s++;
        } else if (a.mode & 4) {   // LDS-DMA, 16 bytes per lane
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(off), "s"(p), "s"(lds0 + (unsigned)(s & 7) * 1024u) : "memory");
        } else {
          double v;
          asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(off), "s"(p) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // plain loads are consumed at once (no prefetch): worst case
          acc += v;
        }
      }
    }
    // stores: streams nl .. nl+nst-1
    if (a.mode & 2) {
      for (int s = 0; s < a.nst; s++) {
        if (lin) { st<BPL>(a.base + linpos, lane * BPL, (double)it); linpos += 1024; }
        else st<BPL>(row + (size_t)(a.nl + s) * a.sstride, off, (double)it);
      }
    }
    switch (a.cap) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
      case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
      case 60: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
      default: break;
    }
  }
  if (acc == 12345.678) a.base[0] = 1;
}

int main(int argc, char **argv) {
  // defaults = the PL pass at 4096^2 x 6: 37 strips of 480 bytes, rows of 33 KB, layers 135 MB apart
  int nl = 0, nst = 6, steps = 30, nstrips = 36, bpl = 8, mode = 2, cap = 30, wpc = 8, rows = 4096, reps = 5;
  size_t pitch = 36864, sstride = (size_t)36864 * 4098;
  for (int i = 1; i < argc; i++) {
    auto val = [&](const char *k) -> const char * { size_t n = strlen(k); return !strncmp(argv[i], k, n) && argv[i][n] == '=' ? argv[i] + n + 1 : nullptr; };
    const char *v;
    if ((v = val("nl"))) nl = atoi(v); else if ((v = val("nst"))) nst = atoi(v); else if ((v = val("steps"))) steps = atoi(v);
    else if ((v = val("nstrips"))) nstrips = atoi(v); else if ((v = val("bpl"))) bpl = atoi(v); else if ((v = val("mode"))) mode = atoi(v);
    else if ((v = val("cap"))) cap = atoi(v); else if ((v = val("wpc"))) wpc = atoi(v); else if ((v = val("rows"))) rows = atoi(v);
    else if ((v = val("pitch"))) pitch = atol(v); else if ((v = val("sstride"))) sstride = atol(v); else if ((v = val("reps"))) reps = atoi(v);
  }
  const int strip_bytes = 64 * bpl - (bpl == 8 ? 32 : 64);   // 60 owned lanes
  if ((size_t)nstrips * 960 + 1024 > pitch) { printf("strips exceed the row\n"); return 1; }
  const int chunks = rows / steps;
  size_t total = sstride * (nl + nst) + (size_t)rows * pitch;
  if (total < (size_t)(rows / steps) * nstrips * steps * (nl + nst) * 1024 + 4096) total = (size_t)(rows / steps) * nstrips * steps * (nl + nst) * 1024 + 4096;
  char *buf;
  CHK(hipMalloc(&buf, total));
  CHK(hipMemset(buf, 0, total));
  Args a{buf, sstride, pitch, nl, nst, steps, nstrips, strip_bytes, mode, cap};
  const size_t lds = 160 * 1024 / wpc - 512;   // dynamic LDS per 64-thread workgroup => wpc wavefronts per CU
  auto kern = bpl == 8 ? k_stream<8> : k_stream<16>;
  CHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int nwaves = chunks * nstrips;
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(kern, dim3(nwaves), dim3(64), lds, 0, a);
  CHK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(nwaves), dim3(64), lds, 0, a);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double ninstr = (double)nwaves * steps * (((mode & 1) ? ((mode & 16) ? nl / 2 : nl) : 0) + ((mode & 2) ? nst : 0));
  const double bytes = (double)nwaves * steps * ((((mode & 1) ? nl : 0) * (double)((mode & 16) ? 512 : (mode & 4) ? 1024 : 64 * 8)) + ((mode & 2) ? nst : 0) * 64.0 * bpl);
  printf("mode=%d nl=%d nst=%d bpl=%d cap=%d wpc=%d steps=%d waves=%d: %.4f ms  %.2f TB/s  %.2f G instr/s\n", mode, nl, nst, bpl, cap, wpc, steps, nwaves, ms,
         bytes / ms * 1e-9, ninstr / ms * 1e-6);
  return 0;
}
