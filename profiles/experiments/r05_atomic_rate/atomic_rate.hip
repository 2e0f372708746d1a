// How fast is the pillariser's histogram pass?  1.44 M rows, one returning atomicAdd each on a table of 1 M cells (random cells, like a
// uniform cloud), against the same pass with a non-returning atomic, with a plain store, and with the rows read only.
// build: hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k_pass(const float *__restrict__ pts, int n, int stride, int *__restrict__ table, int *__restrict__ cell_out,
                                               int *__restrict__ rank_out, int cells) {
  const int base = blockIdx.x * 1024;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int r = base + i * 256 + threadIdx.x;
    if (r < n) {
      const float x = pts[(long long)r * stride + 1], y = pts[(long long)r * stride + 2], b = pts[(long long)r * stride];
      const int cx = (int)floorf((x + 51.2f) / 0.2f), cy = (int)floorf((y + 51.2f) / 0.2f);
      int c = ((int)b * 512 + cx) * 512 + cy;
      if (c < 0 || c >= cells) c = 0;
      cell_out[r] = c;
      if (MODE == 0) rank_out[r] = atomicAdd(&table[c], 1);
      if (MODE == 1) { __hip_atomic_fetch_add(&table[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rank_out[r] = 0; }
      if (MODE == 2) { table[c] = 1; rank_out[r] = 0; }
      if (MODE == 3) rank_out[r] = 0;
      if (MODE == 4) rank_out[r] = __hip_atomic_fetch_add(&table[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 5) {                                         // 32 x 32 transpose inside every 4-KB block of the table: neighbouring cells on different lines
        const int cp = (c & ~0x3ff) | ((c & 31) << 5) | ((c >> 5) & 31);
        rank_out[r] = atomicAdd(&table[cp], 1);
      }
      if (MODE == 6) {                                         // neighbouring cells 64 KB apart (16 k cells): different lines AND different 4-KB pages
        const int cp = (c & ~0x7ffff) | ((c & 31) << 14) | ((c >> 5) & 0x3fff);
        rank_out[r] = atomicAdd(&table[cp], 1);
      }
    }
  }
}

int main(int argc, char **argv) {
  const bool ring = argc > 1 && argv[1][0] == 'r';
  const int frames = 4, n = 1440000, stride = 7, cells = frames * 512 * 512;
  std::vector<float> h((size_t)n * stride);
  srand(1);
  for (int r = 0; r < n; r++) {
    h[(size_t)r * stride] = (float)(r / (n / frames));
    if (ring) {                                                // the LiDAR-like cloud of bench.py --dist ring: r = 70 u^2 around the sensor
      const float u = rand() / (RAND_MAX + 1.0f), th = 6.2831853f * (rand() / (RAND_MAX + 1.0f));
      float x = 70.f * u * u * cosf(th), y = 70.f * u * u * sinf(th);
      if (fabsf(x) >= 51.2f || fabsf(y) >= 51.2f) x = y = 0.f;
      h[(size_t)r * stride + 1] = x;
      h[(size_t)r * stride + 2] = y;
    } else {
      h[(size_t)r * stride + 1] = -51.2f + 102.4f * (rand() / (RAND_MAX + 1.0f));
      h[(size_t)r * stride + 2] = -51.2f + 102.4f * (rand() / (RAND_MAX + 1.0f));
    }
  }
  float *pts; int *table, *cell, *rank;
  hipMalloc(&pts, h.size() * 4); hipMalloc(&table, cells * 4); hipMalloc(&cell, n * 4); hipMalloc(&rank, n * 4);
  hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[7] = {"returning atomicAdd (agent scope)", "non-returning atomic add", "plain store", "rows only", "returning atomic, workgroup scope",
                          "returning atomic, table transposed in 4-KB blocks", "returning atomic, neighbours 64 KB apart"};
  for (int mode = 0; mode < 7; mode++) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
      hipMemsetAsync(table, 0, cells * 4, 0);
      hipEventRecord(e0, 0);
      const int blocks = (n + 1023) / 1024;
      if (mode == 0) hipLaunchKernelGGL(k_pass<0>, dim3(blocks), dim3(256), 0, 0, pts, n, stride, table, cell, rank, cells);
      if (mode == 1) hipLaunchKernelGGL(k_pass<1>, dim3(blocks), dim3(256), 0, 0, pts, n, stride, table, cell, rank, cells);
      if (mode == 2) hipLaunchKernelGGL(k_pass<2>, dim3(blocks), dim3(256), 0, 0, pts, n, stride, table, cell, rank, cells);
      if (mode == 3) hipLaunchKernelGGL(k_pass<3>, dim3(blocks), dim3(256), 0, 0, pts, n, stride, table, cell, rank, cells);
      if (mode == 4) hipLaunchKernelGGL(k_pass<4>, dim3(blocks), dim3(256), 0, 0, pts, n, stride, table, cell, rank, cells);
      if (mode == 5) hipLaunchKernelGGL(k_pass<5>, dim3(blocks), dim3(256), 0, 0, pts, n, stride, table, cell, rank, cells);
      if (mode == 6) hipLaunchKernelGGL(k_pass<6>, dim3(blocks), dim3(256), 0, 0, pts, n, stride, table, cell, rank, cells);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("%-52s %7.1f us\n", names[mode], best * 1e3f);
  }
  return 0;
}
