// pvol_grid.hip -- device build of the photon search structure that replaces
// `volumeMap = new KdTree<Photon>(volumePhotons)` (core/photonshooter.cpp:502-503, core/kdtree.h:100-147).
//
// A uniform grid over the photon bounding box: photons are radix-sorted by cell id (x fastest), so
// every row of cells along x is ONE contiguous range of the sorted SoA arrays and a lookup touches
// (2R+1)^2 ranges with coalesced loads.  The sort is stable, so photons of one cell stay in upload
// order and the structure is bit-reproducible.
//
// Clumpy maps (pinkfloyd: the photon-weighted mean cell occupancy is ~4 700 where the mean is 1.4 -- beams) get a second
// level: inside every cell the photons are ordered by a 4 x 4 x 4 sub-cell id (x fastest again), and `subStart` holds the
// start of every sub-cell.  Rows of cells stay contiguous, so large-radius lookups read the coarse table as before; a lookup
// whose radius is below a cell walks sub-cell rows and tests ~7x fewer photons (pvol_group_dev.h grid_row_range).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "pvol_dev.h"

struct GridBuildArgs {
    const float *p;      // n x 3 (upload order)
    const float *wi;     // n x 3
    const float *alpha;  // n x 30
    uint32_t n;
    float lo[3];
    float inv;
    int32_t gdim[3];
    int32_t sub;         // 1, or 4: sort by (cell, 4x4x4 sub-cell) and fill subStart
    // volume (for the Inside() test of HomogeneousVolumeDensity::p, volumes/homogeneous.h:76-79)
    int32_t volKind;
    float extLo[3], extHi[3];
    float w2v[16];
};

__device__ __forceinline__ int cell_coord(float x, float lo, float inv, int dim) {
    int c = (int)floorf((x - lo) * inv);
    return min(max(c, 0), dim - 1);
}

__global__ void cell_keys_kernel(GridBuildArgs a, uint32_t *keys, uint32_t *vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    int cx = cell_coord(a.p[3 * i], a.lo[0], a.inv, a.gdim[0]);
    int cy = cell_coord(a.p[3 * i + 1], a.lo[1], a.inv, a.gdim[1]);
    int cz = cell_coord(a.p[3 * i + 2], a.lo[2], a.inv, a.gdim[2]);
    uint32_t key = (uint32_t)((cz * a.gdim[1] + cy) * a.gdim[0] + cx);
    if (a.sub > 1) {   // sub-cell RELATIVE to the cell chosen above: the coarse assignment does not depend on the second level
        const float fs = (float)a.sub;
        const int sx = min(max((int)floorf(((a.p[3 * i] - a.lo[0]) * a.inv - (float)cx) * fs), 0), a.sub - 1);
        const int sy = min(max((int)floorf(((a.p[3 * i + 1] - a.lo[1]) * a.inv - (float)cy) * fs), 0), a.sub - 1);
        const int sz = min(max((int)floorf(((a.p[3 * i + 2] - a.lo[2]) * a.inv - (float)cz) * fs), 0), a.sub - 1);
        key = key * (uint32_t)(a.sub * a.sub * a.sub) + (uint32_t)((sz * a.sub + sy) * a.sub + sx);
    }
    keys[i] = key;
    vals[i] = i;
}

// sorted slot j <- photon order[j]; one 8-lane group writes one 128-B alpha row
__global__ void scatter_photons_kernel(GridBuildArgs a, const uint32_t *order, float4 *pos4, float4 *alpha4, float4 *wi4) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t j = gid >> 3, q = gid & 7;
    if (j >= a.n) return;
    uint32_t src = order[j];
    float x = a.p[3 * src], y = a.p[3 * src + 1], z = a.p[3 * src + 2];
    bool inside = true;
    if (a.volKind != PVOL_VOLUME_GRID) {
        const float *m = a.w2v;
        float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
        float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
        float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
        float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
        if (wp != 1.f) { float inv = 1.f / wp; xp *= inv; yp *= inv; zp *= inv; }
        inside = xp >= a.extLo[0] && xp <= a.extHi[0] && yp >= a.extLo[1] && yp <= a.extHi[1] && zp >= a.extLo[2] && zp <= a.extHi[2];
    }
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t bin = 4 * q + c;
        v[c] = (bin < 30 && inside) ? a.alpha[(size_t)src * 30 + bin] : 0.f;
    }
    alpha4[(size_t)j * 8 + q] = make_float4(v[0], v[1], v[2], v[3]);
    if (q == 0) {
        pos4[j] = make_float4(x, y, z, __uint_as_float(src));
        wi4[j] = make_float4(a.wi[3 * src], a.wi[3 * src + 1], a.wi[3 * src + 2], 0.f);
    }
}

// cellStart[c] = first sorted slot whose key >= c * mult (lower bound); cellStart[ncells] = n
__global__ void cell_start_kernel(const uint32_t *sortedKeys, uint32_t n, uint32_t ncells, uint32_t mult, uint32_t *cellStart) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncells) return;
    const unsigned long long want = (unsigned long long)c * mult;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if ((unsigned long long)sortedKeys[mid] < want) lo = mid + 1; else hi = mid;
    }
    cellStart[c] = lo;
}

// sum over cells of (photons in the cell)^2: divided by n, the occupancy of the cell an average PHOTON sits in
__global__ void occupancy_kernel(const uint32_t *cellStart, uint32_t ncells, double *out) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    if (c < ncells) { const double m = (double)(cellStart[c + 1] - cellStart[c]); v = m * m; }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(out, v);
}
extern "C" hipError_t pvol_grid_occupancy(const uint32_t *cellStart, uint32_t ncells, double *sumSquares, hipStream_t stream) {
    double *d = 0;
    hipError_t e = hipMalloc(&d, sizeof(double));
    if (e != hipSuccess) return e;
    hipMemsetAsync(d, 0, sizeof(double), stream);
    hipLaunchKernelGGL(occupancy_kernel, dim3((ncells + 255) / 256), dim3(256), 0, stream, cellStart, ncells, d);
    e = hipMemcpyAsync(sumSquares, d, sizeof(double), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(d);
    return e;
}

// Runs the whole build on `stream`.  All output buffers are preallocated by the caller:
// pos4[n], alpha4[8n], wi4[n], cellStart[ncells+1], and with args->sub == 4 subStart[64 ncells + 1].  Returns a hipError_t.
extern "C" hipError_t pvol_build_grid(const GridBuildArgs *args, float4 *pos4, float4 *alpha4, float4 *wi4,
                                      uint32_t *cellStart, uint32_t *subStart, hipStream_t stream) {
    const uint32_t n = args->n;
    const uint32_t ncells = (uint32_t)args->gdim[0] * args->gdim[1] * args->gdim[2];
    const uint32_t sub3 = args->sub > 1 ? (uint32_t)(args->sub * args->sub * args->sub) : 1u;
    if (sub3 > 1 && (!subStart || (unsigned long long)ncells * sub3 >= 0xffffffffull)) return hipErrorInvalidValue;
    uint32_t *keysIn = 0, *keysOut = 0, *valsIn = 0, *valsOut = 0;
    void *temp = 0;
    size_t tempBytes = 0;
    hipError_t e = hipSuccess;
#define CK(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)
    CK(hipMalloc(&keysIn, sizeof(uint32_t) * n));
    CK(hipMalloc(&keysOut, sizeof(uint32_t) * n));
    CK(hipMalloc(&valsIn, sizeof(uint32_t) * n));
    CK(hipMalloc(&valsOut, sizeof(uint32_t) * n));
    hipLaunchKernelGGL(cell_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, *args, keysIn, valsIn);
    CK(hipGetLastError());
    {
        int bits = 1;
        while ((1ull << bits) < (unsigned long long)ncells * sub3) ++bits;
        CK(hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, keysIn, keysOut, valsIn, valsOut, (int)n, 0, bits, stream));
        CK(hipMalloc(&temp, tempBytes));
        CK(hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, keysIn, keysOut, valsIn, valsOut, (int)n, 0, bits, stream));
    }
    {
        unsigned long long threads = (unsigned long long)n * 8ull;
        hipLaunchKernelGGL(scatter_photons_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, *args, valsOut, pos4, alpha4, wi4);
        CK(hipGetLastError());
    }
    hipLaunchKernelGGL(cell_start_kernel, dim3((ncells + 1 + 255) / 256), dim3(256), 0, stream, keysOut, n, ncells, sub3, cellStart);
    CK(hipGetLastError());
    if (sub3 > 1) {
        const uint32_t nsub = ncells * sub3;
        hipLaunchKernelGGL(cell_start_kernel, dim3((nsub + 1 + 255) / 256), dim3(256), 0, stream, keysOut, n, nsub, 1u, subStart);
        CK(hipGetLastError());
    }
    CK(hipStreamSynchronize(stream));
done:
#undef CK
    if (keysIn) hipFree(keysIn);
    if (keysOut) hipFree(keysOut);
    if (valsIn) hipFree(valsIn);
    if (valsOut) hipFree(valsOut);
    if (temp) hipFree(temp);
    return e;
}
