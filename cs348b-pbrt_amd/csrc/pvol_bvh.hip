// pvol_bvh.hip -- device build of the triangle hierarchy that replaces `new BVHAccel(prims, ...)` (accelerators/bvh.cpp:
// 301-389, core/api.cpp:1284-1290) for scenes with more triangles than the DevScene's embedded array.
//
// A linear BVH: 30-bit Morton codes of the triangle centroids, made unique by the triangle index in the low word and
// radix-sorted (hipCUB, stable); the binary radix tree over the sorted keys is built one inner node per thread (the common-
// prefix search of Karras 2012: direction, range, split); boxes are fitted bottom-up, the second thread to arrive at a
// node continues to its parent.  The reference builds a SAH (or this fork's Morton-AAC) tree on the CPU; what a traversal
// RETURNS does not depend on the tree (pvol_bvh_dev.h: smallest t, highest original index on equal t, padded boxes), so
// the shape is free to be the one a GPU builds in a few kernels: every pass is one read and one write per triangle.
//
// Algorithmic bytes: 36 B read + 8 B key written per triangle (keys), 2 x 8 B per sort pass, 48 B written per leaf and
// 64 B per inner node -- 0.2 ms for 10^5 triangles is launch latency, not bandwidth.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "pvol_dev.h"

struct BvhBuildArgs {
    const float *tri;        // [n][9] world-space vertices, upload order
    const int32_t *mat;      // [n]
    const int32_t *flip;     // [n]
    uint32_t n;
    float pad;               // boxes grow by this much on every side
};

// floats as integers whose order is the floats' order (for atomicMin / atomicMax)
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void bvh_bounds_kernel(BvhBuildArgs a, uint32_t *bounds) {   // bounds[0..2] min, [3..5] max of the centroids
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float *t = a.tri + 9 * (size_t)i;
    for (int k = 0; k < 3; ++k) {
        const float c = (t[k] + t[3 + k] + t[6 + k]) * (1.f / 3.f);
        atomicMin(&bounds[k], f2ord(c));
        atomicMax(&bounds[3 + k], f2ord(c));
    }
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) {   // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__global__ void bvh_keys_kernel(BvhBuildArgs a, const uint32_t *bounds, unsigned long long *keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float *t = a.tri + 9 * (size_t)i;
    uint32_t code = 0;
    for (int k = 0; k < 3; ++k) {
        const float lo = ord2f(bounds[k]), hi = ord2f(bounds[3 + k]);
        const float c = (t[k] + t[3 + k] + t[6 + k]) * (1.f / 3.f);
        const float ext = hi - lo;
        const float u = ext > 0.f ? (c - lo) / ext : 0.f;
        const uint32_t q = (uint32_t)fminf(fmaxf(u * 1024.f, 0.f), 1023.f);
        code |= spread3(q) << (2 - k);
    }
    keys[i] = ((unsigned long long)code << 32) | i;
}

// length of the common prefix of keys i and j, -1 outside the array (the keys are unique: their low words differ)
__device__ __forceinline__ int bvh_delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// one inner node per thread: children and covered range (Karras, "Maximizing parallelism in the construction of BVHs,
// octrees and k-d trees", HPG 2012, section 3).  child >= 0: inner node; < 0: leaf ~child.  parent[] of inner node c at
// parent[c], of leaf s at parent[n - 1 + s].
__global__ void bvh_tree_kernel(const unsigned long long *keys, int n, int2 *children, int2 *range, int *parent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (bvh_delta(keys, n, i, i + 1) - bvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = bvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (bvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (bvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = bvh_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (bvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int left = (lo == gamma) ? ~gamma : gamma;
    const int right = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    children[i] = make_int2(left, right);
    range[i] = make_int2(lo, hi);
    parent[left >= 0 ? left : n - 1 + ~left] = i;
    parent[right >= 0 ? right : n - 1 + ~right] = i;
    if (i == 0) parent[0] = -1;
}

// leaves in sorted order + bottom-up boxes.  box[c] (6 floats) of inner node c, box[n - 1 + s] of leaf s.
__global__ void bvh_fit_kernel(BvhBuildArgs a, const unsigned long long *keys, const int2 *children, const int2 *range, const int *parent,
                               float *box, uint32_t *arrived, float4 *tris, float4 *nodes) {
    const int n = (int)a.n;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t orig = (uint32_t)(keys[s] & 0xffffffffull);
    const float *t = a.tri + 9 * (size_t)orig;
    tris[3 * (size_t)s] = make_float4(t[0], t[1], t[2], __int_as_float((int)orig));
    tris[3 * (size_t)s + 1] = make_float4(t[3], t[4], t[5], __int_as_float(a.mat[orig]));
    tris[3 * (size_t)s + 2] = make_float4(t[6], t[7], t[8], __int_as_float(a.flip[orig]));
    float *b = box + 6 * (size_t)(n - 1 + s);
    for (int k = 0; k < 3; ++k) {
        b[k] = fminf(fminf(t[k], t[3 + k]), t[6 + k]) - a.pad;
        b[3 + k] = fmaxf(fmaxf(t[k], t[3 + k]), t[6 + k]) + a.pad;
    }
    int node = parent[n - 1 + s];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;   // the sibling's subtree is not finished: its thread goes on
        __threadfence();
        const int2 ch = children[node];
        const int2 rg = range[node];
        const float *bl = box + 6 * (size_t)(ch.x >= 0 ? ch.x : n - 1 + ~ch.x);
        const float *br = box + 6 * (size_t)(ch.y >= 0 ? ch.y : n - 1 + ~ch.y);
        float l6[6], r6[6];
        for (int k = 0; k < 6; ++k) {   // written by other threads of this launch: read past the vector cache
            l6[k] = __hip_atomic_load(bl + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r6[k] = __hip_atomic_load(br + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        nodes[4 * (size_t)node] = make_float4(l6[0], l6[1], l6[2], __int_as_float(ch.x));
        nodes[4 * (size_t)node + 1] = make_float4(l6[3], l6[4], l6[5], __int_as_float(ch.y));
        nodes[4 * (size_t)node + 2] = make_float4(r6[0], r6[1], r6[2], __int_as_float(rg.x));
        nodes[4 * (size_t)node + 3] = make_float4(r6[3], r6[4], r6[5], __int_as_float(rg.y));
        float *bo = box + 6 * (size_t)node;
        for (int k = 0; k < 3; ++k) {
            __hip_atomic_store(bo + k, fminf(l6[k], r6[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bo + 3 + k, fmaxf(l6[3 + k], r6[3 + k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        node = parent[node];
    }
}

#define BVH_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = e_; goto done; } } while (0)

// tris: [n][3] float4, nodes: [n-1][4] float4 (device, allocated by the caller).  Synchronous on `stream`.
extern "C" hipError_t pvol_build_bvh(const float *dTri, const int32_t *dMat, const int32_t *dFlip, uint32_t n, float pad, float4 *tris,
                                     float4 *nodes, hipStream_t stream) {
    if (n < 2) return hipErrorInvalidValue;
    hipError_t rc = hipSuccess;
    BvhBuildArgs a;
    a.tri = dTri; a.mat = dMat; a.flip = dFlip; a.n = n; a.pad = pad;
    uint32_t *bounds = 0, *arrived = 0;
    unsigned long long *keys = 0, *keysSorted = 0;
    int2 *children = 0, *range = 0;
    int *parent = 0;
    float *box = 0;
    void *tmp = 0;
    size_t tmpBytes = 0;
    const uint32_t hb[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    const int T = 256;
    const uint32_t B = (n + T - 1) / T;
    BVH_TRY(hipMalloc(&bounds, 6 * 4));
    BVH_TRY(hipMalloc(&keys, (size_t)n * 8));
    BVH_TRY(hipMalloc(&keysSorted, (size_t)n * 8));
    BVH_TRY(hipMalloc(&children, (size_t)n * sizeof(int2)));
    BVH_TRY(hipMalloc(&range, (size_t)n * sizeof(int2)));
    BVH_TRY(hipMalloc(&parent, (size_t)(2 * n) * 4));
    BVH_TRY(hipMalloc(&box, (size_t)(2 * n) * 6 * 4));
    BVH_TRY(hipMalloc(&arrived, (size_t)n * 4));
    BVH_TRY(hipMemcpyAsync(bounds, hb, sizeof(hb), hipMemcpyHostToDevice, stream));
    BVH_TRY(hipMemsetAsync(arrived, 0, (size_t)n * 4, stream));
    hipLaunchKernelGGL(bvh_bounds_kernel, dim3(B), dim3(T), 0, stream, a, bounds);
    hipLaunchKernelGGL(bvh_keys_kernel, dim3(B), dim3(T), 0, stream, a, bounds, keys);
    BVH_TRY(hipcub::DeviceRadixSort::SortKeys(0, tmpBytes, keys, keysSorted, (int)n, 0, 62, stream));
    BVH_TRY(hipMalloc(&tmp, tmpBytes));
    BVH_TRY(hipcub::DeviceRadixSort::SortKeys(tmp, tmpBytes, keys, keysSorted, (int)n, 0, 62, stream));
    hipLaunchKernelGGL(bvh_tree_kernel, dim3(B), dim3(T), 0, stream, keysSorted, (int)n, children, range, parent);
    hipLaunchKernelGGL(bvh_fit_kernel, dim3(B), dim3(T), 0, stream, a, keysSorted, children, range, parent, box, arrived, tris, nodes);
    BVH_TRY(hipGetLastError());
    BVH_TRY(hipStreamSynchronize(stream));
done:
    hipFree(bounds); hipFree(keys); hipFree(keysSorted); hipFree(children); hipFree(range); hipFree(parent); hipFree(box); hipFree(arrived);
    hipFree(tmp);
    return rc;
}
