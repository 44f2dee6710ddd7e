// Probe of the two inline-asm idioms li_group_kernel relies on (v_pk_add_f32 with a negated, broadcast subtrahend; v_cmp + v_addc bit
// insertion): hipcc -O3 --offload-arch=gfx950 tools/asm_probe.hip -o tools/asm_probe.out && ./tools/asm_probe.out  ->  "bad 0" on an MI355X
#include <hip/hip_runtime.h>
typedef float nf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ nf2 pk_sub_lo(nf2 a, nf2 pp) { nf2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(pp)); return r; }
__device__ __forceinline__ nf2 pk_sub_hi(nf2 a, nf2 pp) { nf2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(pp)); return r; }
__device__ __forceinline__ void bit_lt(uint32_t &m, float d2, float th) { asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d2), "v"(th) : "vcc"); }
__global__ void k(const nf2 *a, const nf2 *p, nf2 *o, uint32_t *mo, float th) {
    int i = threadIdx.x;
    nf2 x = pk_sub_lo(a[i], p[i]), y = pk_sub_hi(a[i], p[i]);
    o[2*i] = x; o[2*i+1] = y;
    uint32_t m = 0;
    bit_lt(m, a[i].x, th); bit_lt(m, a[i].y, th); bit_lt(m, p[i].x, th);
    mo[i] = m;
}
int main() {
    nf2 *a, *p, *o; uint32_t *mo;
    hipMallocManaged(&a, 64*8); hipMallocManaged(&p, 64*8); hipMallocManaged(&o, 128*8); hipMallocManaged(&mo, 256);
    for (int i = 0; i < 64; ++i) { a[i] = nf2{1.5f*i, 2.25f*i+1}; p[i] = nf2{0.3f*i, 7.f-i}; }
    hipLaunchKernelGGL(k, 1, 64, 0, 0, a, p, o, mo, 20.f);
    hipDeviceSynchronize();
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        nf2 x = o[2*i], y = o[2*i+1];
        if (x.x != a[i].x - p[i].x || x.y != a[i].y - p[i].x || y.x != a[i].x - p[i].y || y.y != a[i].y - p[i].y) ++bad;
        uint32_t m = ((a[i].x < 20.f) << 2) | ((a[i].y < 20.f) << 1) | (p[i].x < 20.f);
        if (m != mo[i]) ++bad;
    }
    printf("bad %d\n", bad);
    return bad != 0;
}
