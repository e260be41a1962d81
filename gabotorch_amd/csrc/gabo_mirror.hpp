// Symmetric completion shared by the pairwise kernels: the x1-is-x2 paths evaluate and store only i <= j, then this
// kernel copies the upper triangle onto the lower one.  A template only so that each translation unit can carry its own
// copy without an ODR clash.
#pragma once
#include "gabo_device.hpp"

namespace gabo {

// out[j][i] = out[i][j] for i < j, 32x32 tiles transposed through LDS (reads and writes both run along rows)
template <int TAG>
__global__ __launch_bounds__(256) void mirror_upper_kernel(double* __restrict__ out, int64_t n, int tiles) {
    __shared__ double tile[32][33];
    const int64_t b = blockIdx.y;
    // block id -> (ti <= tj) over the upper triangle of the tile grid
    int64_t t = blockIdx.x;
    int ti = 0;
    while (t >= tiles - ti) { t -= tiles - ti; ++ti; }
    int tj = ti + (int)t;
    double* o = out + b * n * n;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int64_t i = (int64_t)ti * 32 + r, j = (int64_t)tj * 32 + tx;
        tile[r][tx] = (i < n && j < n) ? o[i * n + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int64_t j = (int64_t)tj * 32 + r, i = (int64_t)ti * 32 + tx;  // writes row j, columns i
        if (i < n && j < n && i < j) o[j * n + i] = tile[tx][r];
    }
}

// Enumeration of the tiles that touch the upper triangle, shared by the kernels and their launchers: column group cg (of
// `cols` columns) owns row chunks [0, min(row_chunks, ceil((cg+1)*cols/rows))).
__host__ __device__ inline int64_t sym_chunks_of(int64_t cg, int cols, int rows, int64_t row_chunks) {
    int64_t cnt = ((cg + 1) * (int64_t)cols + rows - 1) / rows;
    return cnt < row_chunks ? cnt : row_chunks;
}

}  // namespace gabo
